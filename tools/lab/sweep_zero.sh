R=$GRAFT_REPO_ROOT
V="python3 $R/tools/bench_vox.py --iters 300 --pipelined"
for z in 0 128 256 512 1024; do
export PP_STEP_ZERO=$z
echo "=== zero role workgroups = $z"
for a in "--batch 4" "--batch 1" "--batch 4 --n 200000 --half 100 --P 30000" "--batch 1 --n 200000 --half 100 --P 30000" "--batch 4 --half 60 --P 24000 --N 200" "--batch 4 --step 1.0"; do
$V $a | grep "^batch" | cut -c1-75
done
done
