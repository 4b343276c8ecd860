#!/bin/bash
# tools/lab/roles_c5.sh: k_step's role costs at C5 B=4 by leaving roles out (timing only; results are garbage),
# outputs rotating through 3 buffers (1.3 GB), for block orders 1 (binning first) and 3 (1:3 interleave)
cd "$GRAFT_REPO_ROOT"
export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/skip/libpp_hip.so
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3 --n 200000 --half 100 --P 30000"
for mix in 1 3; do for skip in 0 1 2 4 3 5 6 7; do
  echo -n "mix=$mix skip=$skip (1 tile, 2 order, 4 split): "; PP_STEP_MIX=$mix PP_STEP_SKIP=$skip $V 2>/dev/null | tail -1
done; done
