#!/bin/bash
# tools/lab/run1.sh <tag> <c2|c5> <B>
cd "$GRAFT_REPO_ROOT"
[ -f /tmp/c2.bin ] || python3 tools/lab/gen_pts.py /tmp/c2.bin 4 60000 50
[ -f /tmp/c5.bin ] || python3 tools/lab/gen_pts.py /tmp/c5.bin 4 200000 100
L=tools/lab/_build/$1/vox_lab
if [ "$2" = c2 ]; then $L /tmp/c2.bin $3 60000 50 0.2 12000 100 ${4:-200} ${5:-0}; else $L /tmp/c5.bin $3 200000 100 0.2 30000 100 ${4:-200} ${5:-0}; fi
