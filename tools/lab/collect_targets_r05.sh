#!/bin/bash
# tools/lab/collect_targets_r05.sh: the target-assignment part of tools/collect_profiles_r05.sh alone (after a late change of
# k_targets_gt): kernel stats + PMC, output gpurun_out/prof/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
T="python3 $R/tools/bench_targets.py"
stats() { name=$1; shift; rocprofv3 --kernel-trace --stats -d /tmp/p_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  f=$(find /tmp/p_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
  echo "== $name"; grep -h "us per" $O/$name.log | cut -c1-160; [ -n "$f" ] && grep "pp::" "$f" | cut -d, -f1-4 | head -3; rm -rf /tmp/p_$name; }
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv > /dev/null; echo "== $name"; cat $O/${name}_summary.csv; rm -rf /tmp/c_$name; }
stats targets_c3_b1 $T 250 40 1 2 batch
stats targets_c3_b4 $T 250 40 4 2 batch
stats targets_default_b4 $T 300 40 4 6 batch
stats targets_default_b1 $T 300 40 1 6 batch
pmc pmc_targets_c3_b1_fetch FETCH_SIZE $T 250 40 1 2 batch
pmc pmc_targets_c3_b1_write WRITE_SIZE $T 250 40 1 2 batch
pmc pmc_targets_c3_b4_fetch FETCH_SIZE $T 250 40 4 2 batch
pmc pmc_targets_c3_b4_write WRITE_SIZE $T 250 40 4 2 batch
