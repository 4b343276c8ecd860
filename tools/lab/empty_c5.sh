#!/bin/bash
# tools/lab/empty_c5.sh <tag>...: k_emit at C5 shapes (B=4) on clouds that are entirely out of range (pure zero stores
# in k_emit's pattern) against the real clouds -- is its distance from a plain fill the pattern or the points?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
[ -f /tmp/c5.bin ] || python3 tools/lab/gen_pts.py /tmp/c5.bin 4 200000 100
[ -f /tmp/c5_far.bin ] || python3 - <<PY
import numpy as np
a = np.fromfile('/tmp/c5.bin', np.float32).reshape(4, 200000, 4).copy()
a[..., 0] += 5000.0
a.tofile('/tmp/c5_far.bin')
PY
for tag in "$@"; do for f in c5 c5_far; do
  rm -rf /tmp/e_$f
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/e_$f -o t --output-format csv -- $GRAFT_REPO_ROOT/tools/lab/_build/$tag/vox_lab /tmp/$f.bin 4 200000 100 0.2 30000 100 200 1 > /dev/null 2>&1)
  python3 - $tag $f $(find /tmp/e_$f -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[3])):
    if 'k_emit' in r['Name']:
        print(f"{sys.argv[1]:8s} {sys.argv[2]:7s} k_emit avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:7.2f}")
PY
done; done
