#!/bin/bash
# tools/lab/prefetch_c5.sh: k_step at C5 / reference default B=4, voxelizer-only loop with rotating outputs:
# prefetch workgroups x block order (product library, real results)
cd "$GRAFT_REPO_ROOT"
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3"
for shape in "--n 200000 --half 100 --P 30000" "--half 60 --P 24000 --N 200" ""; do
for mix in 1 3; do for pf in 0 32 128; do
  echo -n "[$shape] mix=$mix prefetch=$pf: "; PP_STEP_MIX=$mix PP_STEP_PREFETCH=$pf $V $shape 2>/dev/null | tail -1 | cut -c1-90
done; done; done
