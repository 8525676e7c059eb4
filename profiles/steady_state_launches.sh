#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/steady_state_launches.sh <tag> [bench flags ...]
# Launches and kernel time of ONE steady-state step: two rocprofv3 --kernel-trace --stats runs of the same bench command with 3 and 8
# timed steps; per kernel (calls_8 - calls_3) / 5 and (ns_8 - ns_3) / 5 — model construction, warm-up and the plan builds cancel.
set -u
TAG=$1; shift
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
for K in 3 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$K -- python3 bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --steps $K --warmup 2 > $OUT/s$K.log 2>&1
done
python3 - <<PY
import csv, glob, json
def load(k):
    f = glob.glob("$OUT/s%d/**/*kernel_stats.csv" % k, recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(3), load(8)
rows = []
for n, (c8, t8) in b.items():
    c3, t3 = a.get(n, (0, 0.0))
    if c8 != c3 or t8 != t3:
        rows.append((n, (c8 - c3) / 5.0, (t8 - t3) / 5e6))
rows.sort(key=lambda r: -r[2])
tot_c, tot_t = sum(r[1] for r in rows), sum(r[2] for r in rows)
out = {"launches_per_step": round(tot_c, 1), "kernel_ms_per_step": round(tot_t, 2),
       "kernels": [{"name": n[:100], "launches": round(c, 1), "ms": round(t, 3)} for n, c, t in rows if c > 0.05]}
json.dump(out, open("$OUT/steady_state.json", "w"), indent=1)
print("launches per step", out["launches_per_step"], "kernel ms per step", out["kernel_ms_per_step"])
for k in out["kernels"][:60]:
    print(f"{k['launches']:8.1f} {k['ms']:9.3f} ms  {k['name'][:90]}")
PY
find $OUT -name "*kernel_trace.csv" -delete
