#!/bin/bash
# tools/lab/sweep_cand.sh: candidates per PAIR workgroup (PP_TARGETS_CAND) x batch, product library
root=$(cd "$(dirname "$0")/../.." && pwd)
for B in 1 4; do for c in ${CANDS:-0 16 20 24 32 48 64}; do
  echo "== B=$B cand=$c"; PP_TARGETS_CAND=$c python $root/tools/bench_targets.py 250 40 $B 2 batch 2>/dev/null | grep anchors=grid | cut -c1-100
done; done
