#!/bin/bash
# tools/lab/knobs_c5.sh: k_step at C5 B=4 (rotating outputs), the tile role's XCD map (PP_STEP_TILE_XCD=0/1; unset: by size)
# x the prefetch role's size, alternating
cd "$GRAFT_REPO_ROOT"
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3 --n 200000 --half 100 --P 30000"
for rep in 1 2 3 4; do for xcd in 0 1; do for pf in ${PFS:-0 64 128}; do
  echo -n "rep=$rep tile_xcd=$xcd prefetch=$pf: "; PP_STEP_TILE_XCD=$xcd PP_STEP_PREFETCH=$pf $V 2>/dev/null | tail -1 | cut -c1-110
done; done; done
