#!/bin/bash
# tools/lab/iou_sq.sh <name> [bench_targets args]: SQ counters of k_targets (batch form only), two --pmc passes
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/c_${name}1 -o sq1 --output-format csv -- python3 $R/tools/bench_targets.py "$@" > $R/gpurun_out/r4/${name}_sq1.log 2>&1 &&
python3 $R/tools/pmc_summary.py /tmp/c_${name}1 $R/gpurun_out/r4/sq1_${name}_summary.csv | grep "k_targets<" &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/c_${name}2 -o sq2 --output-format csv -- python3 $R/tools/bench_targets.py "$@" > $R/gpurun_out/r4/${name}_sq2.log 2>&1 &&
python3 $R/tools/pmc_summary.py /tmp/c_${name}2 $R/gpurun_out/r4/sq2_${name}_summary.csv | grep "k_targets<"
