"""Dev tool: per-step kernel breakdown from a rocprofv3 --kernel-trace CSV.

Steps are delimited by dispatches of a marker kernel (default pp::k_bin_count); the last K
full steps are aggregated by kernel name, and the idle time between dispatches is reported
(what a HIP graph could remove).

usage: step_breakdown.py <kernel_trace.csv> [K] [marker-substring]
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
marker = sys.argv[3] if len(sys.argv) > 3 else "k_bin_count"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < K + 1:
    sys.exit(f"only {len(marks)} marker dispatches")
lo, hi = marks[-K - 1], marks[-1]
sel = rows[lo:hi]
span = rows[hi][0] - rows[lo][0]
busy = defaultdict(float)
calls = defaultdict(int)
gap = 0.0
prev_end = None
for s, e, n in sel:
    busy[n] += e - s
    calls[n] += 1
    if prev_end is not None and s > prev_end:
        gap += s - prev_end
    prev_end = max(prev_end or e, e)
tot = sum(busy.values())
print(f"{K} steps: {span / K / 1e3:.1f} us/step wall, {tot / K / 1e3:.1f} us/step in kernels, "
      f"{gap / K / 1e3:.1f} us/step idle between dispatches, {len(sel) / K:.1f} dispatches/step")
for n, t in sorted(busy.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{t / K / 1e3:9.1f} us/step {calls[n] / K:6.1f} calls/step {t / calls[n] / 1e3:9.1f} us avg  {n[:110]}")
