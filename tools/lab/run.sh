#!/bin/bash
# tools/lab/run.sh <tag> : C2/C5 at B=1/4 through the lab binary of that build
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
[ -f /tmp/c2.bin ] || python3 tools/lab/gen_pts.py /tmp/c2.bin 4 60000 50
[ -f /tmp/c5.bin ] || python3 tools/lab/gen_pts.py /tmp/c5.bin 4 200000 100
L=tools/lab/_build/$1/vox_lab
$L /tmp/c2.bin 1 60000 50 0.2 12000 100
$L /tmp/c2.bin 4 60000 50 0.2 12000 100
$L /tmp/c5.bin 1 200000 100 0.2 30000 100
$L /tmp/c5.bin 4 200000 100 0.2 30000 100
