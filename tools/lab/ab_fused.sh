#!/bin/bash
# tools/lab/ab_fused.sh: the fused one-launch form (k_step<pfn>) under block orders and prefetch widths, alternating
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do for knobs in "" "PP_STEP_MIX=2" "PP_STEP_MIX=3" "PP_STEP_MIX=5" "PP_STEP_PREFETCH=64" "PP_STEP_PREFETCH=256" "PP_STEP_PREFETCH=0"; do
  echo -n "[$knobs] "; env $knobs python3 tools/bench_fused_vox.py 4 2>/dev/null | grep pipelined | cut -c50-110
done; done
