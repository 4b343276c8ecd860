#!/bin/bash
# tools/lab/prof1.sh <tag> <c2|c5> <B> [order]: rocprofv3 kernel stats of one lab run
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
[ -f /tmp/c2.bin ] || python3 tools/lab/gen_pts.py /tmp/c2.bin 4 60000 50
[ -f /tmp/c5.bin ] || python3 tools/lab/gen_pts.py /tmp/c5.bin 4 200000 100
L=$GRAFT_REPO_ROOT/tools/lab/_build/$1/vox_lab
name=$1_$2b$3o${4:-0}
if [ "$2" = c2 ]; then args="/tmp/c2.bin $3 60000 50 0.2 12000 100 200 ${4:-0}"; else args="/tmp/c5.bin $3 200000 100 0.2 30000 100 200 ${4:-0}"; fi
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name --output-format csv -- $L $args > /tmp/prof_$name.log 2>&1)
grep "us/step" /tmp/prof_$name.log
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/lab/${name}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if 'pp::' in r['Name']:
        a=float(r['AverageNs'])/1e3; tot+=a
        print(f"   {r['Name'][:28]:30s} avg={a:8.2f} us  min={float(r['MinNs'])/1e3:.2f} max={float(r['MaxNs'])/1e3:.2f}")
print(f"   sum of kernels {tot:.2f} us")
PY
