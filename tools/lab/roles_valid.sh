#!/bin/bash
# tools/lab/roles_valid.sh: k_step's roles left out AFTER 40 complete launches -- the later roles then work on the
# real lists of the same cloud (valid timing of the emit role alone, of emit + order, ...), C2 and C5, B=4
cd "$GRAFT_REPO_ROOT"
export PP_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/_build/skip/libpp_hip.so PP_STEP_SKIP_AFTER=40
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3"
for shape in "" "--n 200000 --half 100 --P 30000"; do
  echo -n "[$shape] three launches: "; python3 tools/bench_vox.py --batch 4 --iters 150 --rotate 3 $shape 2>/dev/null | tail -1 | cut -c40-140
  for mix in 1 3; do for skip in 0 4 5 7; do for pf in 128 0; do
  echo -n "[$shape] mix=$mix skip=$skip prefetch=$pf: "; PP_STEP_PREFETCH=$pf PP_STEP_MIX=$mix PP_STEP_SKIP=$skip $V $shape 2>/dev/null | tail -1 | cut -c40-140
done; done; done; done
