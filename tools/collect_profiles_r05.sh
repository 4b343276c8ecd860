#!/bin/bash
# tools/collect_profiles_r05.sh: the rocprofv3 evidence of round 5 (run on the GPU box through gpurun; output: gpurun_out/prof/,
# copied to profiles/r05/).  Kernel stats (--kernel-trace --stats) and counters in SEPARATE passes (--pmc with --kernel-trace
# only), from /tmp, the program directly after `--`.  What VERDICT r4 asked for by name: targets_c3_b1_*, vox_c2_b1_*,
# fused_c2_b4_* (stats + PMC + SQ); plus the driver's bench command, its headline loop, the stress / reference-default /
# row-major shapes and the batched target assignment.
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
RD="--half 60 --P 24000 --N 200"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
# the driver's command and its headline loop alone (k_step's average there covers exactly the launches `roofline` times)
stats bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin
stats bench_headline python3 $R/bench.py --steps 50 --warmup 10 --headline-only
pmc pmc_bench_headline_fetch FETCH_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
pmc pmc_bench_headline_write WRITE_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
# target assignment: ONE sample per launch (configs[3]'s per-GPU shape) and the batch of a step
stats targets_c3_b1 $T 250 40 1 2 batch
pmc pmc_targets_c3_b1_fetch FETCH_SIZE $T 250 40 1 2 batch
pmc pmc_targets_c3_b1_write WRITE_SIZE $T 250 40 1 2 batch
pmc sq1_targets_c3_b1 "$SQ1" $T 250 40 1 2 batch
pmc sq2_targets_c3_b1 "$SQ2" $T 250 40 1 2 batch
stats targets_c3_b4 $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_fetch FETCH_SIZE $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_write WRITE_SIZE $T 250 40 4 2 batch
pmc sq1_targets_c3_b4 "$SQ1" $T 250 40 4 2 batch
stats targets_default_b4 $T 300 40 4 6 batch
stats targets_default_b1 $T 300 40 1 6 batch
# the voxelizer at ONE sweep per launch (outputs into 12 buffers in turn) and at four (4 buffers in turn)
stats vox_c2_b1_step $V --batch 1 --pipelined --rotate 12
stats vox_c2_b1_three $V --batch 1 --rotate 12
pmc pmc_c2_b1_step_fetch FETCH_SIZE $V --batch 1 --iters 50 --pipelined --rotate 12
pmc pmc_c2_b1_step_write WRITE_SIZE $V --batch 1 --iters 50 --pipelined --rotate 12
pmc sq1_c2_b1_step "$SQ1" $V --batch 1 --iters 50 --pipelined --rotate 12
pmc sq2_c2_b1_step "$SQ2" $V --batch 1 --iters 50 --pipelined --rotate 12
stats vox_c2_b4_step_rotate $V --batch 4 --pipelined --rotate 4
stats vox_c2_b4_three_rotate $V --batch 4 --rotate 4
stats vox_c2_b4_rowmajor_step $V --batch 4 --order 0 --pipelined --rotate 4
stats vox_c2_b1_rowmajor_step $V --batch 1 --order 0 --pipelined --rotate 12
stats vox_c5_b4_step $V --batch 4 $C5 --pipelined --rotate 2
stats vox_c5_b1_step $V --batch 1 $C5 --pipelined --rotate 6
stats vox_refdef_b4_step $V --batch 4 $RD --pipelined --rotate 2
stats vox_refdef_b1_step $V --batch 1 $RD --pipelined --rotate 4
pmc pmc_c5_b4_step_fetch FETCH_SIZE $V --batch 4 $C5 --iters 50 --pipelined --rotate 2
pmc pmc_c5_b4_step_write WRITE_SIZE $V --batch 4 $C5 --iters 50 --pipelined --rotate 2
# the fused feature-net call: three launches and the one-launch form (k_step<3, 0>)
stats fused_c2_b4 $F 4
stats fused_c2_b1 $F 1
pmc pmc_fused_c2_b4_fetch FETCH_SIZE $F 4
pmc pmc_fused_c2_b4_write WRITE_SIZE $F 4
pmc sq1_fused_c2_b4 "$SQ1" $F 4
pmc sq2_fused_c2_b4 "$SQ2" $F 4
ls $O | wc -l
