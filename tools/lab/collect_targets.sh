#!/bin/bash
# tools/lab/collect_targets.sh: the target-assignment part of tools/collect_profiles.sh alone (output: gpurun_out/prof/)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  f=$(find /tmp/p_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
  echo "== $name"; [ -n "$f" ] && grep "pp::" "$f" | cut -d, -f1-4 | head -4
}
pmc() { name=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv > /dev/null
  echo "== $name"; grep "k_targets" $O/${name}_summary.csv
}
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
T="python3 $R/tools/bench_targets.py"
stats targets_c3 $T 250 40 4 2 single
stats targets_c3_b4 $T 250 40 4 2 batch
stats targets_default_b4 $T 300 40 4 6 batch
PP_TARGETS_FORM=anchors stats targets_c3_b4_anchor_form $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_fetch FETCH_SIZE $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_write WRITE_SIZE $T 250 40 4 2 batch
pmc sq1_targets_c3_b4 "$SQ1" $T 250 40 4 2 batch
pmc sq2_targets_c3_b4 "$SQ2" $T 250 40 4 2 batch
