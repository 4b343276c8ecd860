#!/bin/bash
# tools/lab/iou_time.sh <tag>...: target-assignment wall time with each variant library
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do echo -n "$t: "; PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/$t/libpp_hip.so python3 tools/bench_targets.py 2>/dev/null | tail -1; done
