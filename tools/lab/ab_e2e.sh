#!/bin/bash
# tools/lab/ab_e2e.sh <tag>: the end-to-end headline (bench.py --headline-only), product library against a
# build_full.sh variant, alternating within one call
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do for v in product $1; do
  if [ $v = product ]; then unset PP_HIP_LIB; else export PP_HIP_LIB=$R/tools/lab/_build/$v/libpp_hip.so; fi
  echo -n "$v: "; python3 bench.py --steps 50 --warmup 10 --headline-only 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'sweeps/s', round(j['ms_per_step'],4), 'ms/step')"
done; done
