#!/bin/bash
# tools/lab/sweep_rowmajor.sh: tile count (PP_TARGET_TILES) for the row-major pillar order and for C1's 100x100 grid, k_step
cd "$GRAFT_REPO_ROOT"
for t in 0 245 489 977 1953; do
  if [ $t = 0 ]; then unset PP_TARGET_TILES; else export PP_TARGET_TILES=$t; fi
  for shape in "--batch 4 --order 0" "--batch 1 --order 0" "--batch 4 --step 1.0" "--batch 4"; do
    echo "tiles=$t [$shape]: $(python3 tools/bench_vox.py --pipelined --iters 200 --rotate 4 $shape 2>/dev/null | tail -1 | cut -c38-110)"
  done
done
