#!/bin/bash
# tools/lab/sweep_sub.sh: k_step's sub-batch launches (PP_STEP_SUB_MB = dense bytes per launch; 0 = one launch per call)
cd "$GRAFT_REPO_ROOT"
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150"
for mb in 0 128 256 512; do
  export PP_STEP_SUB_MB=$mb
  echo "== PP_STEP_SUB_MB=$mb"
  echo -n "C2      : "; $V 2>/dev/null | tail -1
  echo -n "C5      : "; $V --n 200000 --half 100 --P 30000 2>/dev/null | tail -1
  echo -n "refdef  : "; $V --half 60 --P 24000 --N 200 2>/dev/null | tail -1
done
