#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/collect_r02.sh <tag>
# Round 2: the bench line is the full default invocation (headline + reference-YAML shapes + LoRA/Muon config); the profiled
# runs add --no-extras so that the trace holds the headline configuration only.
# Produces under gpurun_out/<tag>/ : the bench JSON line, the rocprofv3 kernel-trace stats of the same command, and two
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots") for the roofline
# kernel's HBM traffic.  No --pmc pass is combined with any tracing other than --kernel-trace.
set -u
TAG=${1:-r02_a}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $OUT/bench.json.log 2> $OUT/bench.err.log
tail -1 $OUT/bench.json.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/pmc_write.log 2>&1
# BASELINE configs[2] (LoRA r16 + Muon + stochastic depth + deep SpecAugment, B = 32): kernel-trace stats of its own run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lora_stats -- python3 bench.py --lora --muon --stochastic-depth 0.1 --deep-spec-augment --batch 32 --no-cpu-baseline --no-roofline --no-extras --steps 5 --warmup 2 > $OUT/lora_stats.log 2>&1
python3 - <<PY
import csv, glob, json, collections
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "$OUT/pmc_fetch" if name == "FETCH_SIZE" else "$OUT/pmc_write"
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
    out[name] = {k: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for k, v in acc.items() if "gemm" in k or "attn" in k}
out["batch"] = 68
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps({k: {kk: round(vv["mean"]) for kk, vv in v.items()} for k, v in out.items() if k != "batch"})[:1500])
PY
ls $OUT/stats/*/ | head
# the raw per-dispatch traces are tens of MiB each: gpurun only copies back 64 MiB, keep the summaries
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
