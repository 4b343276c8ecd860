#!/bin/bash
# tools/pipe_exp.sh: plain vs software-pipelined (k_step) voxelizer loop at C2/C5, B=1/4 (run through gpurun)
set -u
R=$GRAFT_REPO_ROOT
V="python3 $R/tools/bench_vox.py --iters 300"
for b in 1 4; do
  echo "== C2 B=$b plain"; $V --batch $b
  echo "== C2 B=$b pipelined"; $V --batch $b --pipelined
done
for b in 1 4; do
echo "== C5 B=$b plain"; $V --batch $b --n 200000 --half 100 --P 30000
echo "== C5 B=$b pipelined"; $V --batch $b --n 200000 --half 100 --P 30000 --pipelined
done
echo "== C2 B=4 row-major plain"; $V --batch 4 --order 0
echo "== C2 B=4 row-major pipelined"; $V --batch 4 --order 0 --pipelined
echo "== C1 B=1 plain"; $V --batch 1 --step 1.0
echo "== C1 B=1 pipelined"; $V --batch 1 --step 1.0 --pipelined
