#!/bin/bash
# tools/lab/emit_series.sh <tag> <file> <iters>: k_emit's duration launch by launch (C5 shapes, B=4)
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/es
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/es -o t --output-format csv -- $GRAFT_REPO_ROOT/tools/lab/_build/$1/vox_lab /tmp/$2.bin 4 200000 100 0.2 30000 100 ${3:-200} 1 > /dev/null 2>&1)
python3 - $(find /tmp/es -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
d = [(e - s) / 1e3 for s, e, n in rows if 'k_emit' in n]
t = [s for s, e, n in rows if 'k_emit' in n]
print(len(d), 'launches; durations us:')
for i in range(0, len(d), 20):
    print(f"{i:4d} @{(t[i]-t[0])/1e6:7.2f} ms: " + " ".join(f"{x:5.1f}" for x in d[i:i+20]))
PY
