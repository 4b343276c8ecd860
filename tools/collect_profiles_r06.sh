#!/bin/bash
# tools/collect_profiles_r06.sh: the rocprofv3 evidence of round 6 (run on the GPU box through gpurun; output: gpurun_out/prof/,
# copied to profiles/r06/).  Kernel stats (--kernel-trace --stats) and counters in SEPARATE passes (--pmc with --kernel-trace
# only), from /tmp, the program directly after `--`.  What VERDICT r5 asked for by name: vox_c1_* (BASELINE config 1's grid),
# pmc_c5_b4_step_*; plus the driver's bench command, its headline loop, the target assignment and the host drop-in's calls.
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
  echo "== $name"; grep -h "us/step\|us per\|\"metric\"" $O/$name.log | cut -c1-200
  [ -n "$f" ] && grep "pp::" "$f" | cut -d, -f1-4 | head -8
  rm -rf /tmp/p_$name
}
pmc() { # name "counters" cmd...
  name=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv > /dev/null
  echo "== $name"; cat $O/${name}_summary.csv
  rm -rf /tmp/c_$name
}
V="python3 $R/tools/bench_vox.py"
T="python3 $R/tools/bench_targets.py"
F="python3 $R/tools/bench_fused_vox.py"
C5="--n 200000 --half 100 --P 30000"
C1="--step 1.0"
RD="--half 60 --P 24000 --N 200"
# the driver's command and its headline loop alone (k_step's average there covers exactly the launches `roofline` times)
stats bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-live-traffic
stats bench_headline python3 $R/bench.py --steps 50 --warmup 10 --headline-only
pmc pmc_bench_headline_fetch FETCH_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
pmc pmc_bench_headline_write WRITE_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
# BASELINE config 1's 100 x 100 grid on the GPU: the streamed path of crowded waves
stats vox_c1_b4_step $V --batch 4 $C1 --pipelined --rotate 4
stats vox_c1_b1_step $V --batch 1 $C1 --pipelined --rotate 12
stats vox_c1_b4_three $V --batch 4 $C1 --rotate 4
stats vox_c1_b4_rowmajor_three $V --batch 4 $C1 --order 0 --rotate 4
pmc pmc_c1_b4_step_fetch FETCH_SIZE $V --batch 4 $C1 --iters 50 --pipelined --rotate 4
pmc pmc_c1_b4_step_write WRITE_SIZE $V --batch 4 $C1 --iters 50 --pipelined --rotate 4
# target assignment: ONE sample per launch (configs[3]'s per-GPU shape) and the batch of a step
stats targets_c3_b1 $T 250 40 1 2 batch
stats targets_c3_b4 $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_fetch FETCH_SIZE $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_write WRITE_SIZE $T 250 40 4 2 batch
# the voxelizer at ONE sweep per launch (outputs into 12 buffers in turn) and at four (4 buffers in turn)
stats vox_c2_b1_step $V --batch 1 --pipelined --rotate 12
stats vox_c2_b4_step_rotate $V --batch 4 --pipelined --rotate 4
stats vox_c2_b4_three_rotate $V --batch 4 --rotate 4
stats vox_c5_b4_step $V --batch 4 $C5 --pipelined --rotate 2
stats vox_c5_b4_three $V --batch 4 $C5 --rotate 2
stats vox_refdef_b4_step $V --batch 4 $RD --pipelined --rotate 2
pmc pmc_c5_b4_step_fetch FETCH_SIZE $V --batch 4 $C5 --iters 50 --pipelined --rotate 2
pmc pmc_c5_b4_step_write WRITE_SIZE $V --batch 4 $C5 --iters 50 --pipelined --rotate 2
pmc pmc_c5_b4_three_fetch FETCH_SIZE $V --batch 4 $C5 --iters 50 --rotate 2
pmc pmc_c5_b4_three_write WRITE_SIZE $V --batch 4 $C5 --iters 50 --rotate 2
# the fused feature-net call (k_step<3, 0>) and the host drop-in's two calls (kernels + copies of one call)
stats fused_c2_b4 $F 4
stats dropin_calls python3 $R/tools/lab/dropin_loop.py 40 both
ls $O | wc -l
