#!/bin/bash
# tools/lab/sweep.sh <tag>: wall time per step over the k_tile knobs
cd "$GRAFT_REPO_ROOT"
for cfg in "c2 1" "c2 4" "c5 1" "c5 4"; do
  for tw in 4 8 16; do for tk in 0 1; do
    echo -n "tw=$tw ticket=$tk  "; PP_TILE_WAVES=$tw PP_FORCE_TICKET=$tk tools/lab/run1.sh $1 $cfg | head -1
  done; done
done
