#!/bin/bash
# tools/lab/ab_b1.sh <tag> ...: k_step at ONE sweep per launch (configs[3]'s per-GPU shape), 12 output buffers in turn,
# library variants alternating within one call; C2 and C5 shapes
root=$(cd "$(dirname "$0")/../.." && pwd)
for rep in 1 2 3; do for tag in "$@"; do
  lib=$root/tools/lab/_build/$tag/libpp_hip.so; [ "$tag" = product ] && lib=$root/3d-object-detection_amd/libpp_hip.so
  for shape in "--batch 1" "--batch 1 --n 200000 --half 100 --P 30000"; do
    echo "== $tag rep $rep $shape: $(PP_HIP_LIB=$lib python $root/tools/bench_vox.py --iters 300 --rotate 12 --pipelined $shape 2>/dev/null | grep -E "^batch|k_step" | tr '\n' ' ' | cut -c1-200)"
  done
done; done
