"""tools/lab/gt_stamps.py: phase stamps of one k_targets_gt launch (library built with -DPP_IOU_STAMPS)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd
from pp_amd import boxes, synth, _lib
from pp_amd.targets import TargetAssigner
fm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ta = TargetAssigner(boxes.AnchorConfig(fm, fm), canvas_height=2 * fm)
counts, packed = ta.upload_batch([synth.gt_boxes(G, 2 * fm, s) for s in range(B)])
for _ in range(20):
    ta.assign_batch_device(counts, packed)
torch.cuda.synchronize()
buf = np.zeros(16 * 4096, np.uint64)
f = _lib.lib().pp_debug_iou_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 16).astype(np.int64)
ok = st[:, 0] > 0
st = st[ok]
t0 = st[:, 0].min()
zero = st[:, 1] < t0          # ZERO-role workgroups never stamp slot 1 in this launch
names = {0: "start", 1: "box loaded", 2: "gated", 4: "clipped", 5: "clipped again (lab)", 6: "reduced", 8: "drained", 9: "ticket", 11: "tail: loads in, LDS armed",
         12: "tail: maxima", 13: "tail: winners", 14: "tail: positives down, forced rows fetched", 15: "positives done", 10: "end"}
print(f"{len(st)} workgroups stamped ({zero.sum()} ZERO role); us relative to the first start: min / median / max")
for k, nm in names.items():
    col = st[:, k]
    m = col >= t0
    if m.any():
        c = (col[m] - t0) / 100.0
        print(f"  {nm:15s} n={m.sum():4d}  {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f}")
p = ~zero
for a, b in ((0, 1), (1, 2), (2, 4), (4, 5), (4, 6), (6, 8), (8, 9)):
    m = p & (st[:, a] >= t0) & (st[:, b] >= st[:, a])
    if not m.any():
        continue
    d = (st[m, b] - st[m, a]) / 100.0
    print(f"  PAIR role {names[a]:>11s} -> {names[b]:11s} n={m.sum():4d} median {np.median(d):5.2f} max {d.max():5.2f}")
z = zero & (st[:, 9] >= t0)
print("  ZERO role lifetime median %.2f max %.2f" % (np.median((st[z, 9] - st[z, 0]) / 100.0), ((st[z, 9] - st[z, 0]) / 100.0).max()))
last = st[st[:, 10] >= t0]
for row in last:
    print("  tail workgroup: " + ", ".join(f"{names[k].split(':')[-1].strip()} {(row[k] - t0) / 100.0:.2f}" for k in (0, 9, 11, 12, 13, 14, 10) if row[k] >= t0))
ends = np.sort((st[p, 9] - t0) / 100.0)
print("  PAIR tickets (us): deciles", np.round(np.percentile(ends, [0, 10, 25, 50, 75, 90, 95, 99, 100]), 2))
starts = np.sort((st[p, 0] - t0) / 100.0)
print("  PAIR starts (us): deciles", np.round(np.percentile(starts, [0, 10, 25, 50, 75, 90, 95, 99, 100]), 2))
