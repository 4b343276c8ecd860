#!/bin/bash
# tools/lab/ab_c1.sh <tagA> <tagB>: BASELINE config 1's 100 x 100 grid (and C2 / C5 / reference default as regression checks)
# with two library variants alternating inside ONE call: k_step pipelined and the three-launch k_emit, both orders, B = 4 / 1
a=$1; b=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
run() { # tag args...
  tag=$1; shift
  lib=$root/tools/lab/_build/$tag/libpp_hip.so; [ "$tag" = product ] && lib=$root/3d-object-detection_amd/libpp_hip.so
  PP_HIP_LIB=$lib python $root/tools/bench_vox.py "$@" 2>/dev/null | tail -2 | tr '\n' ' '; echo
}
for rep in 1 2; do
  for order in 1 0; do
    for B in 4 1; do
      for tag in $a $b; do
        echo "== C1 order=$order B=$B $tag (rep $rep) pipelined"; run $tag --step 1.0 --batch $B --order $order --rotate 4 --pipelined
        echo "== C1 order=$order B=$B $tag (rep $rep) three-launch"; run $tag --step 1.0 --batch $B --order $order --rotate 4
      done
    done
  done
done
for tag in $a $b $a $b; do
  echo "== C2 B=4 $tag pipelined"; run $tag --batch 4 --rotate 4 --pipelined
  echo "== C2 B=1 $tag pipelined"; run $tag --batch 1 --rotate 12 --pipelined
  echo "== C5 B=4 $tag pipelined"; run $tag --n 200000 --half 100 --P 30000 --batch 4 --rotate 2 --pipelined --iters 100
  echo "== refdef B=4 $tag pipelined"; run $tag --half 60 --P 24000 --N 200 --batch 4 --rotate 2 --pipelined --iters 100
done
