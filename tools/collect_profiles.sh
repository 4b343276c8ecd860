#!/bin/bash
# tools/collect_profiles.sh: the rocprofv3 evidence of one round (run on the GPU box through gpurun):
# kernel stats of the driver's bench command and of the voxelizer-only loop at C2/C5, B=1/4, both
# pillar orders; FETCH_SIZE / WRITE_SIZE in separate --pmc passes.  Output: gpurun_out/prof/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { # name cmd...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  f=$(find /tmp/p_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
  echo "== $name"; grep -h "us/step\|\"metric\"" $O/$name.log | cut -c1-200
  [ -n "$f" ] && grep "pp::" "$f" | cut -d, -f1-4 | head -8
}
pmc() { # name counter cmd...
  name=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv
}
V="python3 $R/tools/bench_vox.py"
stats bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
for o in 1 0; do
  stats vox_c2_b1_o$o $V --batch 1 --order $o
  stats vox_c2_b4_o$o $V --batch 4 --order $o
  stats vox_c5_b1_o$o $V --batch 1 --n 200000 --half 100 --P 30000 --order $o
  stats vox_c5_b4_o$o $V --batch 4 --n 200000 --half 100 --P 30000 --order $o
done
pmc pmc_c2_b4_fetch FETCH_SIZE $V --batch 4 --iters 50
pmc pmc_c2_b4_write WRITE_SIZE $V --batch 4 --iters 50
pmc pmc_c2_b1_fetch FETCH_SIZE $V --batch 1 --iters 50
pmc pmc_c2_b1_write WRITE_SIZE $V --batch 1 --iters 50
pmc pmc_c5_b4_fetch FETCH_SIZE $V --batch 4 --n 200000 --half 100 --P 30000 --iters 50
pmc pmc_c5_b4_write WRITE_SIZE $V --batch 4 --n 200000 --half 100 --P 30000 --iters 50
T="python3 $R/tools/bench_targets.py"
stats targets_c3 $T
pmc pmc_targets_c3_fetch FETCH_SIZE $T
pmc pmc_targets_c3_write WRITE_SIZE $T
