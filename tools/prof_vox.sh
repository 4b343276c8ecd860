#!/bin/bash
# rocprofv3 kernel stats of the voxelizer-only loop: tools/prof_vox.sh <outdir> <tag> <bench_vox args...>
out=$1; tag=$2; shift 2
mkdir -p "$GRAFT_REPO_ROOT/$out"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/$tag" -o "$tag" --output-format csv -- python3 "$GRAFT_REPO_ROOT/tools/bench_vox.py" "$@" > "$GRAFT_REPO_ROOT/$out/$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/$tag" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
tail -1 "$out/$tag.log"
[ -n "$f" ] && cut -d, -f1-4 "$f" | head -8
