#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/collect_r06.sh <tag> [what]
#   what = all (default) | bench | stats | pmc | lora | mfma | base
# Round 6 (headline batch 96).  Produces under gpurun_out/<tag>/ :
#   bench.json.log            the default bench.py invocation (headline + extras: reference-YAML shapes, configs[1] / [2] / [4],
#                             entrypoint loop)
#   stats/                    rocprofv3 --kernel-trace --stats of the headline configuration (--no-extras)
#   pmc_summary.json          FETCH_SIZE / WRITE_SIZE per kernel of the headline (separate --pmc passes)  -> roofline.traffic
#   lora_stats/, lora_pmc_summary.json   the same two for BASELINE configs[2] (LoRA r16 + Muon + SD + deep SpecAugment, B = 32)
#   base_stats/               rocprofv3 --kernel-trace --stats of BASELINE configs[1] (whisper-base, 8 clips, training.wft_hip_graph)
#   mfma_busy.json            SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY,
#                             SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_LDS_BANK_CONFLICT per kernel (headline), for the MFMA-pipe
#                             utilisation of the GEMM and attention kernels (north_star: "rocprof MFMA-utilisation counters")
# Every --pmc pass is its own run with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots"; gpurun refuses --pmc next
# to any other tracing); the program follows `--` directly.
set -u
TAG=${1:-r06_a}
WHAT=${2:-all}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
HEAD="--no-cpu-baseline --no-extras"
LORA="--lora --muon --stochastic-depth 0.1 --deep-spec-augment --batch 32 --no-cpu-baseline --no-extras"
want() { [ "$WHAT" = all ] || [ "$WHAT" = "$1" ]; }

if want bench; then
  python3 bench.py > $OUT/bench.json.log 2> $OUT/bench.err.log
  tail -1 $OUT/bench.json.log | cut -c1-400
fi
if want stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 2 $HEAD > $OUT/stats.log 2>&1
fi
pmc_pass() {  # $1 = output dir, $2 = counters (one pass), rest = bench flags
  local d=$1 c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 1 --no-roofline "$@" > $d.log 2>&1
}
if want pmc; then
  pmc_pass $OUT/pmc_fetch FETCH_SIZE $HEAD
  pmc_pass $OUT/pmc_write WRITE_SIZE $HEAD
fi
if want lora; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lora_stats -- python3 bench.py $LORA --no-roofline --steps 5 --warmup 2 > $OUT/lora_stats.log 2>&1
  pmc_pass $OUT/lora_pmc_fetch FETCH_SIZE $LORA
  pmc_pass $OUT/lora_pmc_write WRITE_SIZE $LORA
fi
if want base; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/base_stats -- python3 bench.py --model base --batch 8 --hip-graph --no-extras --no-cpu-baseline --no-roofline --steps 50 --warmup 10 > $OUT/base_stats.log 2>&1
  tail -1 $OUT/base_stats.log | cut -c1-300
fi
if want mfma; then
  pmc_pass $OUT/pmc_sq1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" $HEAD
  pmc_pass $OUT/pmc_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16" $HEAD
  pmc_pass $OUT/pmc_grbm "GRBM_GUI_ACTIVE" $HEAD
fi
python3 - <<PY
import collections, csv, glob, json, os
OUT = "$OUT"
def collect(dirs_counters, batch, keep=("gemm", "attn", "ln_", "lora", "mt_adamw", "tn_splitk")):
    out = {}
    for d, names in dirs_counters:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(OUT, d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] in names:
                    acc[row["Counter_Name"]][row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
        for cn, per in acc.items():
            out[cn] = {k: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for k, v in per.items() if any(t in k for t in keep)}
    out["batch"] = batch
    return out
def dump(name, obj):
    if len(obj) > 1:
        json.dump(obj, open(os.path.join(OUT, name), "w"), indent=1)
        print(name, {k: len(v) for k, v in obj.items() if k != "batch"})
dump("pmc_summary.json", collect([("pmc_fetch", ("FETCH_SIZE",)), ("pmc_write", ("WRITE_SIZE",))], 96))
dump("lora_pmc_summary.json", collect([("lora_pmc_fetch", ("FETCH_SIZE",)), ("lora_pmc_write", ("WRITE_SIZE",))], 32))
sq = collect([("pmc_sq1", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")),
              ("pmc_sq2", ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_MFMA_MOPS_BF16")),
              ("pmc_grbm", ("GRBM_GUI_ACTIVE",))], 96)
if len(sq) > 1:
    # MFMA-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs
    # x 1024 SIMDs); wave-cycle split WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES
    derived = {}
    gui = sq.get("GRBM_GUI_ACTIVE", {})
    for k, v in sq.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).items():
        row = {"launches": v["launches"]}
        if k in gui and gui[k]["mean"] > 0:
            row["mfma_pipe_busy_frac"] = round(v["mean"] / (gui[k]["mean"] / 8.0 * 1024.0), 4)
        wc = sq.get("SQ_WAVE_CYCLES", {}).get(k)
        for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if wc and k in sq.get(cn, {}) and wc["mean"] > 0:
                row[cn.lower() + "_frac_of_wave_cycles"] = round(sq[cn][k]["mean"] / wc["mean"], 4)
        if k in sq.get("SQ_LDS_BANK_CONFLICT", {}):
            row["lds_bank_conflict_cycles_per_launch"] = round(sq["SQ_LDS_BANK_CONFLICT"][k]["mean"])
        derived[k] = row
    sq["derived"] = derived
    dump("mfma_busy.json", sq)
    print(json.dumps(derived, indent=0)[:3000])
PY
ls $OUT/stats/*/ 2>/dev/null | head -3
# the raw per-dispatch traces are tens of MiB each: gpurun only copies back 64 MiB, keep the summaries
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
