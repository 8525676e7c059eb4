#!/bin/bash
# GPU box: whole-step A/B of library builds under THIS tree's Python (alternating): bash tools/dev/step_lib_ab.sh <rounds> <lib> <lib> ...
R=${1:-3}; shift
for i in $(seq 1 $R); do
  for l in "$@"; do
    WFT_LIB=$(realpath $l) python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['ms_per_step'], d.get('ms_per_step_median'), 'nt256', d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
  done
done
