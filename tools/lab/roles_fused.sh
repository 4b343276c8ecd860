#!/bin/bash
# tools/lab/roles_fused.sh: role costs of the fused one-launch form (k_step<3, 0>: split | tile | order | fused emit | clear)
# by leaving roles out after 40 complete launches (timing only; `skip` build: -DPP_STEP_SKIP_KNOB).  PP_STEP_SKIP bits:
# 1 tile, 2 order, 4 split, 8 clear, 16 emit
cd "$GRAFT_REPO_ROOT"
export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/skip/libpp_hip.so PP_STEP_SKIP_AFTER=40
for B in 4 1; do for rep in 1 2; do
  for skip in 0 8 1 2 4 7 15 16 24 23; do for pf in 128 0; do
    echo -n "B=$B skip=$skip prefetch=$pf: "; PP_STEP_PREFETCH=$pf PP_STEP_SKIP=$skip python3 tools/bench_fused_vox.py $B 2>/dev/null | grep pipelined | cut -c40-120
  done; done
done; done
