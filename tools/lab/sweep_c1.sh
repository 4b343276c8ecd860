R=$GRAFT_REPO_ROOT
V="python3 $R/tools/bench_vox.py --iters 200"
for a in "--batch 4 --step 1.0" "--batch 1 --step 1.0" "--batch 4 --step 0.5" "--batch 4"; do
echo "== $a"; $V $a | grep "kernels" | cut -c1-150; $V $a --pipelined | grep "kernels" | cut -c1-150
done
