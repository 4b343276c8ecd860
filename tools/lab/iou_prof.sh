#!/bin/bash
# tools/lab/iou_prof.sh <tag> [fm G]: per-kernel times of the target assignment with the variant library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab/iou_$1
export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/$1/libpp_hip.so
python3 tools/bench_targets.py ${2:-250} ${3:-40}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lab/iou_$1 -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_targets.py ${2:-250} ${3:-40} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/lab/iou_$1/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_targets' in r['Name']: print(r['Name'][:40], r['Calls'], 'avg %.2f us min %.2f'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
