#!/bin/bash
# tools/lab/pmc_c5.sh [tag]: HBM counters of configs[4]'s voxelizer per kernel -- the three launches (k_split, k_tile,
# k_emit) and k_step -- FETCH_SIZE and WRITE_SIZE in separate passes, medians per launch in KiB
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_c5${1:+_$1}
mkdir -p $O
[ -n "${1:-}" ] && [ "$1" != product ] && export PP_HIP_LIB=$R/tools/lab/_build/$1/libpp_hip.so
cd /tmp && export TMPDIR=/tmp
V="python3 $R/tools/bench_vox.py --batch 4 --n 200000 --half 100 --P 30000 --iters 40 --rotate 2"
for ctr in FETCH_SIZE WRITE_SIZE; do
  for form in three step; do
    extra=""; [ $form = step ] && extra="--pipelined"
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_${form}_$ctr -o t --output-format csv -- $V $extra > $O/${form}_$ctr.log 2>&1
    python3 $R/tools/pmc_summary.py /tmp/c_${form}_$ctr $O/${form}_${ctr}_summary.csv > /dev/null
    echo "== $form $ctr"; cat $O/${form}_${ctr}_summary.csv
    rm -rf /tmp/c_${form}_$ctr
  done
done
