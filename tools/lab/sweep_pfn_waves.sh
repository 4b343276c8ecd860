R=$GRAFT_REPO_ROOT
for v in base pfn7 pfn8; do
  if [ $v = base ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  echo "== $v"; python3 $R/tools/bench_fused_vox.py 4 | grep "reuse=True"; python3 $R/tools/bench_fused_vox.py 1 | grep "reuse=True"
done
