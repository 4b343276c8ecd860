#!/bin/bash
# tools/lab/prefetch_sets.sh: which of the prefetch role's arrays pay for themselves (PP_STEP_PREFETCH_SETS mask:
# 1 emit role's, 2 order role's, 4 tile role's; PP_STEP_PREFETCH_POINTS: the split role's points)
R=$GRAFT_REPO_ROOT
cd $R
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3"
for rep in 1 2 3; do for sets in "7 1" "1 0" "1 1" "5 0" "3 0"; do
  set -- $sets
  export PP_STEP_PREFETCH_SETS=$1 PP_STEP_PREFETCH_POINTS=$2
  c2=$($V 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  c5=$($V --n 200000 --half 100 --P 30000 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  c53=$(PP_STEP_MIX=3 $V --n 200000 --half 100 --P 30000 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  rd=$($V --half 60 --P 24000 --N 200 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  b1=$(python3 tools/bench_vox.py --pipelined --batch 1 --iters 300 --rotate 8 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  hl=$(python3 bench.py --steps 50 --warmup 10 --headline-only 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['roofline']['avg_launch_us'],1), round(j['value'],1))")
  echo "sets=$1 points=$2: C2 $c2  C2/B=1 $b1  C5 $c5  C5(mix 3) $c53  refdef $rd us/step; headline k_step us, sweeps/s $hl"
done; done
