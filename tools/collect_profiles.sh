#!/bin/bash
# tools/collect_profiles.sh: the rocprofv3 evidence of one round (round 4: + batched target assignment, rotating
# output buffers, cache counters of k_step, the fused path's one-launch form) (run on the GPU box through gpurun):
# kernel stats of the driver's bench command and of the voxelizer-only loop at C2/C5, B=1/4, software-pipelined
# (k_step) and as three launches; FETCH_SIZE / WRITE_SIZE in separate --pmc passes; SQ counters of the binning
# kernels.  Output: gpurun_out/prof/
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
pmc() { # name "counters" cmd...
  name=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv > /dev/null
  echo "== $name"; cat $O/${name}_summary.csv
}
V="python3 $R/tools/bench_vox.py"
C5="--n 200000 --half 100 --P 30000"
stats bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
# the headline loop alone: k_step's average here covers exactly the launches bench.py's `roofline` times
stats bench_headline python3 $R/bench.py --steps 50 --warmup 10 --headline-only
pmc pmc_bench_headline_fetch FETCH_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
pmc pmc_bench_headline_write WRITE_SIZE python3 $R/bench.py --steps 20 --warmup 5 --headline-only
for b in 1 4; do
  stats vox_c2_b${b}_step $V --batch $b --pipelined
  stats vox_c2_b${b}_three $V --batch $b
  stats vox_c5_b${b}_step $V --batch $b $C5 --pipelined
  stats vox_c5_b${b}_three $V --batch $b $C5
done
stats vox_c2_b4_rowmajor_step $V --batch 4 --order 0 --pipelined
stats vox_c2_b4_rowmajor_three $V --batch 4 --order 0
for b in 1 4; do
  pmc pmc_c2_b${b}_step_fetch FETCH_SIZE $V --batch $b --iters 50 --pipelined
  pmc pmc_c2_b${b}_step_write WRITE_SIZE $V --batch $b --iters 50 --pipelined
  pmc pmc_c2_b${b}_three_fetch FETCH_SIZE $V --batch $b --iters 50
  pmc pmc_c2_b${b}_three_write WRITE_SIZE $V --batch $b --iters 50
done
pmc pmc_c5_b4_three_fetch FETCH_SIZE $V --batch 4 $C5 --iters 50
pmc pmc_c5_b4_three_write WRITE_SIZE $V --batch 4 $C5 --iters 50
pmc pmc_c5_b1_step_fetch FETCH_SIZE $V --batch 1 $C5 --iters 50 --pipelined
pmc pmc_c5_b1_step_write WRITE_SIZE $V --batch 1 $C5 --iters 50 --pipelined
# where the binning kernels' time goes, by counter (SQ block: 8 per pass; quad-cycle units)
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
for b in 1 4; do
  pmc sq1_c2_b${b}_three "$SQ1" $V --batch $b --iters 50
  pmc sq2_c2_b${b}_three "$SQ2" $V --batch $b --iters 50
  pmc sq1_c2_b${b}_step "$SQ1" $V --batch $b --iters 50 --pipelined
done
pmc sq1_c5_b4_three "$SQ1" $V --batch 4 $C5 --iters 50
pmc sq2_c5_b4_three "$SQ2" $V --batch 4 $C5 --iters 50
T="python3 $R/tools/bench_targets.py"   # (tools/lab/collect_targets.sh runs this part alone)
stats targets_c3 $T 250 40 4 2 single          # one sample per launch (rounds 1-3's form)
stats targets_c3_b4 $T 250 40 4 2 batch         # the batch of a step in one launch
stats targets_default_b4 $T 300 40 4 6 batch    # the reference's shipped anchor set (540 000 anchors)
pmc pmc_targets_c3_b4_fetch FETCH_SIZE $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_write WRITE_SIZE $T 250 40 4 2 batch
pmc sq1_targets_c3_b4 "$SQ1" $T 250 40 4 2 batch
pmc sq2_targets_c3_b4 "$SQ2" $T 250 40 4 2 batch
# k_step with its outputs rotating through >= 512 MB of buffers (real HBM traffic) against the one-buffer loop
stats vox_c2_b4_step_rotate $V --batch 4 --pipelined --rotate 4
stats vox_c2_b4_three_rotate $V --batch 4 --rotate 4
pmc pmc_c2_b4_step_rotate_fetch FETCH_SIZE $V --batch 4 --iters 50 --pipelined --rotate 4
pmc pmc_c2_b4_step_rotate_write WRITE_SIZE $V --batch 4 --iters 50 --pipelined --rotate 4
# cache counters of k_step: the end-to-end loop (the network's activations between two launches), the
# voxelizer-only loop with rotating outputs, and the one-buffer loop
TCC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
pmc tcc_bench_headline "$TCC" python3 $R/bench.py --steps 20 --warmup 5 --headline-only
pmc tcc_c2_b4_step_rotate "$TCC" $V --batch 4 --iters 50 --pipelined --rotate 4
pmc tcc_c2_b4_step_onebuf "$TCC" $V --batch 4 --iters 50 --pipelined
# the fused feature-net call: three launches and the one-launch form
stats fused_c2_b4 python3 $R/tools/bench_fused_vox.py 4
