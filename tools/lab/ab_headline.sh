R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
for m in "" "--three-launch"; do
python3 $R/bench.py --steps 50 --warmup 10 --headline-only $m 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$m'.ljust(15), round(j['value'],1), round(j['ms_per_step'],4), j['roofline']['kernel'][:12], round(j['roofline']['avg_launch_us'],1))"
done; done
