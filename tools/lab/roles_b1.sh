#!/bin/bash
# tools/lab/roles_b1.sh: k_step's role costs at ONE sweep per launch (leaving roles out after 40 complete launches; timing
# only), C2 and C5 shapes, with and without the prefetch role; the early-zero instance beside the single-pass one
cd "$GRAFT_REPO_ROOT"
export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/skip/libpp_hip.so PP_STEP_SKIP_AFTER=40
V="python3 tools/bench_vox.py --pipelined --batch 1 --iters 300 --rotate 12"
for shape in "" "--n 200000 --half 100 --P 30000"; do
  for early in 0 1; do for skip in 0 1 2 4 3 5 6 7; do for pf in 128 0; do
  echo -n "[$shape] early=$early skip=$skip (1 tile, 2 order, 4 split left out) prefetch=$pf: "; PP_EMIT_EARLY=$early PP_STEP_PREFETCH=$pf PP_STEP_SKIP=$skip $V $shape 2>/dev/null | tail -1 | cut -c38-130
done; done; done; done
