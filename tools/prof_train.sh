#!/bin/bash
# tools/prof_train.sh <outdir>: kernel trace of the training step loop + per-step breakdown
out=$1; mkdir -p "$GRAFT_REPO_ROOT/$out"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/tr" -o tr --output-format csv -- python3 "$GRAFT_REPO_ROOT/tools/bench_train.py" > "$GRAFT_REPO_ROOT/$out/train.log" 2>&1
cd "$GRAFT_REPO_ROOT"
tail -1 $out/train.log
f=$(find "$out/tr" -name "*kernel_trace.csv" | head -1)
python3 tools/step_breakdown.py "$f" 8 k_split > $out/train_breakdown.txt
head -45 $out/train_breakdown.txt | cut -c1-170
