set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/c_$name -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/c_$name $O/${name}_summary.csv
}
T="python3 $R/tools/bench_targets.py"
V="python3 $R/tools/bench_vox.py"
pmc pmc_targets_c3_fetch FETCH_SIZE $T
pmc pmc_targets_c3_write WRITE_SIZE $T
pmc pmc_c2_b1_fetch FETCH_SIZE $V --batch 1 --iters 50
pmc pmc_c2_b1_write WRITE_SIZE $V --batch 1 --iters 50
