#!/bin/bash
# tools/lab/mix_b1.sh: k_step's block order at ONE sweep per launch (configs[3] / configs[4] per-GPU shapes), outputs rotating
cd "$GRAFT_REPO_ROOT"
V="python3 tools/bench_vox.py --pipelined --batch 1 --iters 300 --rotate 12"
for rep in 1 2; do for mix in 1 2 3 5; do
  c2=$(PP_STEP_MIX=$mix $V 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  c5=$(PP_STEP_MIX=$mix $V --n 200000 --half 100 --P 30000 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  rd=$(PP_STEP_MIX=$mix $V --half 60 --P 24000 --N 200 2>/dev/null | tail -1 | sed -E 's/.*: +([0-9.]+) us\/step.*/\1/')
  echo "mix=$mix: C2/B=1 $c2  C5/B=1 $c5  refdef/B=1 $rd us/step"
done; done
