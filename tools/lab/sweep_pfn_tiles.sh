# fused one-launch form (k_step<pfn>): tile size x waves-per-SIMD bound, within one call
R=$GRAFT_REPO_ROOT
for v in base spfn6; do
  if [ $v = base ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  for t in 0 256 512; do
    if [ $t = 0 ]; then unset PP_TARGET_TILES; else export PP_TARGET_TILES=$t; fi
    echo "== $v tiles=$t"; python3 $R/tools/bench_fused_vox.py 4 | grep pipelined
  done
done
