#!/bin/bash
# GPU box: whole-step A/B of environment settings (alternating): bash tools/dev/step_env_ab.sh <rounds> "VAR=a" "VAR=b" ...
R=${1:-3}; shift
for i in $(seq 1 $R); do
  for e in "$@"; do
    env $e python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d.get('ms_per_step_median'), 'nt256', d['roofline']['achieved'])"
  done
done
