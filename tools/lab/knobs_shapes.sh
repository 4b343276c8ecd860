#!/bin/bash
# tools/lab/knobs_shapes.sh: k_step's tile-role XCD map (PP_STEP_TILE_XCD) and prefetch size at every shape, alternating
cd "$GRAFT_REPO_ROOT"
run() { python3 "$@" 2>/dev/null | tail -1 | cut -c1-100; }
for rep in 1 2; do for knobs in "0 128" "1 128" "1 256" "0 256"; do set -- $knobs
  export PP_STEP_TILE_XCD=$1 PP_STEP_PREFETCH=$2
  echo -n "rep=$rep xcd=$1 pf=$2 C2 B=4: "; run tools/bench_vox.py --batch 4 --rotate 4 --pipelined
  echo -n "rep=$rep xcd=$1 pf=$2 C2 B=4 cold: "; run tools/bench_vox.py --batch 4 --rotate 4 --pipelined --cold 1024
  echo -n "rep=$rep xcd=$1 pf=$2 C2 B=1: "; run tools/bench_vox.py --batch 1 --rotate 12 --pipelined
  echo -n "rep=$rep xcd=$1 pf=$2 C2 B=4 row-major: "; run tools/bench_vox.py --batch 4 --rotate 4 --pipelined --order 0
  echo -n "rep=$rep xcd=$1 pf=$2 C5 B=4: "; run tools/bench_vox.py --n 200000 --half 100 --P 30000 --batch 4 --rotate 3 --pipelined --iters 150
  echo -n "rep=$rep xcd=$1 pf=$2 C5 B=1: "; run tools/bench_vox.py --n 200000 --half 100 --P 30000 --batch 1 --rotate 8 --pipelined --iters 150
  echo -n "rep=$rep xcd=$1 pf=$2 refdef B=4: "; run tools/bench_vox.py --half 60 --P 24000 --N 200 --batch 4 --rotate 2 --pipelined --iters 100
  echo -n "rep=$rep xcd=$1 pf=$2 C1 B=4: "; run tools/bench_vox.py --step 1.0 --batch 4 --rotate 4 --pipelined --iters 100
done; done
