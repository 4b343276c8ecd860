R=$GRAFT_REPO_ROOT
for pf in 128 0 64 256; do
export PP_STEP_PREFETCH=$pf
python3 $R/bench.py --steps 50 --warmup 10 --headline-only 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('prefetch $pf e2e', round(j['value'],1), round(j['roofline']['avg_launch_us'],1))"
python3 $R/tools/bench_vox.py --iters 300 --batch 4 --pipelined | grep "^batch" | cut -c1-70
done
