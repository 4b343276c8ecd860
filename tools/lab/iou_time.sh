#!/bin/bash
# tools/lab/iou_time.sh <tag>...: target-assignment wall time with each variant library ("base" = the product library)
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  if [ $t = base ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/$t/libpp_hip.so; fi
  echo "== $t"; python3 tools/bench_targets.py 250 40 4 2>/dev/null | grep grid; python3 tools/bench_targets.py 300 40 4 6 2>/dev/null | grep grid
done
