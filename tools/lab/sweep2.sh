#!/bin/bash
# tools/lab/sweep2.sh <tag> <order> "<tiles...>" "<waves...>": target tiles x waves
cd "$GRAFT_REPO_ROOT"
for cfg in "c2 1" "c2 4" "c5 1" "c5 4"; do
  for tt in $3; do for tw in $4; do
    echo -n "o=$2 tiles=$tt tw=$tw  "; PP_TARGET_TILES=$tt PP_TILE_WAVES=$tw tools/lab/run1.sh $1 $cfg 200 $2 | head -1 | cut -c1-75
  done; done
done
