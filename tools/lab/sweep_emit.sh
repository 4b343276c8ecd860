#!/bin/bash
# tools/lab/sweep_emit.sh <tag>: wall time per step over the persistent k_emit grid sizes
cd "$GRAFT_REPO_ROOT"
for cfg in "c2 1" "c2 4" "c5 1" "c5 4"; do
  for wg in 0 768 1024 1280 2048; do
    echo -n "$cfg emit_wgs=$wg  "; PP_EMIT_WGS=$wg tools/lab/run1.sh $1 $cfg | head -2 | tr '\n' ' '; echo
  done
done
