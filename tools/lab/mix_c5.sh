#!/bin/bash
# tools/lab/mix_c5.sh: k_step's block order at C5 B=4 (3400 binning blocks, more than the chip holds at once), with and
# without 1 GiB of unrelated traffic between the calls; twice, to see the noise
cd "$GRAFT_REPO_ROOT"
V="python3 tools/bench_vox.py --pipelined --batch 4 --iters 150 --rotate 3 --n 200000 --half 100 --P 30000"
for rep in 1 2; do for cold in 0 1024; do for mix in 1 2 3 4; do
  echo -n "cold=$cold mix=$mix: "; PP_STEP_MIX=$mix $V --cold $cold 2>/dev/null | tail -1 | cut -c40-140
done; done; done
