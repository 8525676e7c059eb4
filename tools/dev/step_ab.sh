#!/bin/bash
# GPU box: whole-step A/B of this tree against the round-2 tree (_ab_old/, built by `git archive 678076d | tar -x -C _ab_old` +
# make) on ONE box, alternating runs: bash tools/dev/step_ab.sh [rounds] [extra bench flags]
R=${1:-3}; shift
for i in $(seq 1 $R); do
  for t in _ab_old .; do
    (cd $t && python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d.get('ms_per_step_median'))")
  done
done
