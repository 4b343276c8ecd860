R=$GRAFT_REPO_ROOT
V="python3 $R/tools/bench_vox.py --iters 300"
for a in "--batch 4" "--batch 1" "--batch 4 --n 200000 --half 100 --P 30000" "--batch 1 --n 200000 --half 100 --P 30000" "--batch 4 --order 0" "--batch 4 --step 1.0" "--batch 4 --half 60 --P 24000 --N 200"; do
echo "== $a"; $V $a | grep "kernels\|^batch" | cut -c1-150; $V $a --pipelined | grep "kernels\|^batch" | cut -c1-150
done
python3 $R/tools/bench_fused_vox.py 4 | grep fused
python3 $R/tools/step_roles.py --batch 4; python3 $R/tools/step_roles.py --batch 1
