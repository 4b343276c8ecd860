#!/bin/bash
# tools/lab/mix_shapes.sh: k_step's block order (PP_STEP_MIX = m: groups of one binning block and m-1 emit blocks; 1 = every
# binning block first; unset = the library's automatic choice) per shape at B=4, with 1 GiB of unrelated traffic between
# the calls (a network-sized gap, the real caller) and without; kernel medians from the event pairs
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for shape in "--half 60 --P 24000 --N 200|refdef" "|C2" "--n 200000 --half 100 --P 30000|C5"; do
  args=${shape%|*}; name=${shape#*|}
  for cold in 1024 0; do
    line="$name cold=$cold:"
    for mix in auto 1 2 3 4 5 8; do
      if [ $mix = auto ]; then unset PP_STEP_MIX; else export PP_STEP_MIX=$mix; fi
      us=$(python3 tools/bench_vox.py --pipelined --batch 4 --iters 120 --rotate 4 $args --cold $cold 2>/dev/null | tail -1 | sed -E 's/.*emit median ([0-9.]+) us.*/\1/')
      line="$line  mix=$mix $us"
    done
    unset PP_STEP_MIX
    echo "$line"
  done
done; done
