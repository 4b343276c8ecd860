cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pf -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_fused.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pf/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pp::' in r['Name']:
        print(f"{r['Name'][:60]:62s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
