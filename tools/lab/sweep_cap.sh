#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in base cap64w6; do
  if [ $v = base ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  echo "=== $v"
  for args in "--batch 4 --order 0" "--batch 4 --order 0 --pipelined" "--batch 1 --n 200000 --half 100 --P 30000" "--batch 1 --n 200000 --half 100 --P 30000 --pipelined" "--batch 4 --half 60 --P 24000 --N 200" "--batch 4 --half 60 --P 24000 --N 200 --pipelined" "--batch 1 --half 60 --P 24000 --N 200 --pipelined" "--batch 4 --n 200000 --half 100 --P 30000" "--batch 4 --n 200000 --half 100 --P 30000"; do
    echo "-- $args"; python3 $R/tools/bench_vox.py --iters 200 $args 2>&1 | grep "kernels" | cut -c1-150
  done
done
