#!/bin/bash
# tools/lab/ab_lib.sh <tag>: the product library against tools/lab/_build/<tag>/libpp_hip.so within ONE call (boxes differ
# by 5-8 %): k_step in the voxelizer-only loop (outputs in turn) at C2 / C5 / reference default, and inside
# bench.py --headline-only (a network pass between the launches); alternating, three times
R=$GRAFT_REPO_ROOT
cd $R
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3"
for i in 1 2 3; do for v in product $1; do
  if [ $v = product ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  c2=$($V 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  c5=$($V --n 200000 --half 100 --P 30000 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  rd=$($V --half 60 --P 24000 --N 200 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  hl=$(python3 bench.py --steps 50 --warmup 10 --headline-only 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['roofline']['avg_launch_us'],1), round(j['value'],1))")
  echo "$v: C2 $c2  C5 $c5  refdef $rd us/step; headline k_step us, sweeps/s: $hl"
done; done
