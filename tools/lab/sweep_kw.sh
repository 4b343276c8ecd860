R=$GRAFT_REPO_ROOT
for v in base kw3 kw5 kw6; do
  if [ $v = base ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  echo "=== $v"
  timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_voxelize.py -x -q -k "full_size or pipelined or odd_shapes" 2>&1 | tail -1
  for a in "--batch 4" "--batch 1" "--batch 4 --n 200000 --half 100 --P 30000" "--batch 1 --n 200000 --half 100 --P 30000" "--batch 4 --half 60 --P 24000 --N 200"; do
    python3 $R/tools/bench_vox.py --iters 200 $a 2>&1 | grep "kernels" | cut -c1-70; python3 $R/tools/bench_vox.py --iters 200 $a --pipelined 2>&1 | grep "^batch" | cut -c1-75
  done
done
