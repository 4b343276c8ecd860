#!/bin/bash
# tools/lab/ab_early.sh: the early-zero emit instance (PP_EMIT_EARLY=0/1) at one and two sweeps per launch, outputs in turn
root=$(cd "$(dirname "$0")/../.." && pwd)
for rep in 1 2 3; do for e in 0 1; do
  for shape in "--batch 1" "--batch 2" "--batch 1 --n 200000 --half 100 --P 30000" "--batch 1 --order 0"; do
    echo "== early=$e rep $rep $shape: $(PP_EMIT_EARLY=$e python $root/tools/bench_vox.py --iters 300 --rotate 12 --pipelined $shape 2>/dev/null | grep -E "^batch" | cut -c1-120)"
  done
done; done
