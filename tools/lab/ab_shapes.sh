#!/bin/bash
# tools/lab/ab_shapes.sh <tag>...: regression check of k_step at C2 / C5 / the reference default and of the fused one-launch
# form, library variants alternating inside ONE call
root=$(cd "$(dirname "$0")/../.." && pwd)
run() { tag=$1; shift
  lib=$root/tools/lab/_build/$tag/libpp_hip.so; [ "$tag" = product ] && lib=$root/3d-object-detection_amd/libpp_hip.so
  PP_HIP_LIB=$lib python "$@" 2>/dev/null | tail -1 | cut -c1-150; }
for rep in 1 2; do for tag in "$@"; do
  echo -n "C2 B=4 $tag: "; run $tag $root/tools/bench_vox.py --batch 4 --rotate 4 --pipelined
  echo -n "C2 B=1 $tag: "; run $tag $root/tools/bench_vox.py --batch 1 --rotate 12 --pipelined
  echo -n "C2 B=4 row-major $tag: "; run $tag $root/tools/bench_vox.py --batch 4 --rotate 4 --pipelined --order 0
  echo -n "C5 B=4 $tag: "; run $tag $root/tools/bench_vox.py --n 200000 --half 100 --P 30000 --batch 4 --rotate 2 --pipelined --iters 100
  echo -n "C5 B=4 row-major $tag: "; run $tag $root/tools/bench_vox.py --n 200000 --half 100 --P 30000 --batch 4 --rotate 2 --pipelined --iters 100 --order 0
  echo -n "refdef B=4 $tag: "; run $tag $root/tools/bench_vox.py --half 60 --P 24000 --N 200 --batch 4 --rotate 2 --pipelined --iters 100
  echo -n "fused B=4 $tag: "; run $tag $root/tools/bench_fused_vox.py 4
done; done
