#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
V="python3 $R/tools/bench_vox.py --iters 300 --pipelined"
$V --batch 1 | grep "kernels\|^batch" | cut -c1-140; $V --batch 4 | grep "kernels\|^batch" | cut -c1-140
$V --batch 4 --cold 1024 | grep "kernels" | cut -c1-140
$V --batch 1 --n 200000 --half 100 --P 30000 | grep "kernels\|^batch"| cut -c1-140; $V --batch 4 --n 200000 --half 100 --P 30000 | grep "kernels\|^batch"| cut -c1-140
$V --batch 4 --order 0 | grep "kernels\|^batch"| cut -c1-140
python3 $R/tools/step_roles.py --batch 4; python3 $R/tools/step_roles.py --batch 1
