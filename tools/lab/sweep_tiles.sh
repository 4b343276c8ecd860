#!/bin/bash
# tools/lab/sweep_tiles.sh <tag>: wall time and per-kernel times over the split's tile count (scrambled order)
cd "$GRAFT_REPO_ROOT"
for cfg in "c2 1" "c2 4" "c5 1" "c5 4"; do
  for tt in 64 128 256 512 1024; do
    echo -n "$cfg tiles<=$tt  "; PP_TARGET_TILES=$tt tools/lab/run1.sh $1 $cfg 200 1 | head -3 | tr '\n' ' ' | cut -c1-330; echo
  done
done
