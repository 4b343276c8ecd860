#!/bin/bash
# tools/lab/ab_c1_short.sh <tag>...: BASELINE config 1's 100 x 100 grid, library variants alternating inside ONE call
root=$(cd "$(dirname "$0")/../.." && pwd)
run() { tag=$1; shift
  lib=$root/tools/lab/_build/$tag/libpp_hip.so; [ "$tag" = product ] && lib=$root/3d-object-detection_amd/libpp_hip.so
  PP_HIP_LIB=$lib python $root/tools/bench_vox.py "$@" 2>/dev/null | tail -2 | tr '\n' ' ' | sed 's/host:.*us per call until the device is done//; s/-> .*//'; echo; }
for rep in 1 2; do for order in 1 0; do for B in 4 1; do for tag in "$@"; do
  echo -n "C1 order=$order B=$B $tag pipelined: "; run $tag --step 1.0 --batch $B --order $order --rotate 4 --pipelined
  echo -n "C1 order=$order B=$B $tag three-launch: "; run $tag --step 1.0 --batch $B --order $order --rotate 4
done; done; done; done
