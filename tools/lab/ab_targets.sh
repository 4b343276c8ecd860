#!/bin/bash
# tools/lab/ab_targets.sh <tagA> <tagB> [fm G per_cell]: target assignment with two library variants, alternating within
# ONE call (boxes differ by 5-15 %): one sample per launch and a batch of four, then the stamps of <tag>_stamps builds
a=$1; b=$2; fm=${3:-250}; G=${4:-40}; pc=${5:-2}
root=$(cd "$(dirname "$0")/../.." && pwd)
for rep in 1 2 3; do
  for tag in $a $b; do
    lib=$root/tools/lab/_build/$tag/libpp_hip.so; [ "$tag" = product ] && lib=$root/3d-object-detection_amd/libpp_hip.so
    echo "== $tag (rep $rep)"
    PP_HIP_LIB=$lib python $root/tools/bench_targets.py $fm $G 4 $pc grid 2>/dev/null | grep anchors=grid
  done
done
for tag in $a $b; do
  lib=$root/tools/lab/_build/${tag}_stamps/libpp_hip.so; [ "$tag" = product ] && lib=$root/tools/lab/_build/new_stamps/libpp_hip.so
  [ -f $lib ] || continue
  for B in 1 4; do
    echo "== stamps $tag B=$B"
    PP_HIP_LIB=$lib python $root/tools/lab/gt_stamps.py $fm $G $B 2>/dev/null
  done
done
