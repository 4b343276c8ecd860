"""tools/lab/iou_stamps.py: phase stamps of one k_targets launch (library built with -DPP_IOU_STAMPS)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd
from pp_amd import boxes, synth, _lib
from pp_amd.targets import TargetAssigner
fm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = boxes.AnchorConfig(fm, fm)
gt = synth.gt_boxes(G, 2 * fm, 0)
ta = TargetAssigner(cfg, canvas_height=2 * fm)
counts, packed = ta.upload_batch([synth.gt_boxes(G, 2 * fm, s) for s in range(B)])
for _ in range(20):
    ta.assign_batch_device(counts, packed)
torch.cuda.synchronize()
nwg = min(4096, B * ((ta.A + 255) // 256))
buf = np.zeros(16 * 4096, np.uint64)
f = _lib.lib().pp_debug_iou_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 16)[:nwg].astype(np.int64)
t0 = st[:, 0].min()
names = ["start", "staged", "gated", "queued", "clipped", "reduced", "rows out", "appended", "drained", "ticket", "tail", "t:loaded", "t:argmax", "t:forced?", "t:rows0", "t:dups"]
print(f"{nwg} workgroups; 10 ns ticks relative to the first start; per stamp: min / median / max over the workgroups that have it")
for k, nm in enumerate(names):
    col = st[:, k]
    ok = col >= t0
    if ok.any():
        c = (col[ok] - t0) / 100.0
        print(f"  {nm:9s} n={ok.sum():4d}  {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f} us")
dur = (st[:, 9] - st[:, 0]) / 100.0
for a, b_ in ((0, 1), (1, 2), (2, 5), (5, 6), (6, 8), (8, 9)):
    ok = (st[:, a] >= t0) & (st[:, b_] >= t0)
    d = (st[ok, b_] - st[ok, a]) / 100.0
    print(f"  {names[a]:>9s} -> {names[b_]:9s} n={ok.sum():4d} median {np.median(d):6.2f} mean {d.mean():6.2f} max {d.max():6.2f} us")
print("workgroup lifetime: median %.2f max %.2f us; clip phase (queued->clipped) median %.2f max %.2f us" % (
    np.median(dur), dur.max(), np.median((st[:, 4] - st[:, 3])[st[:, 3] >= t0]) / 100.0, ((st[:, 4] - st[:, 3])[st[:, 3] >= t0]).max() / 100.0))
# timeline: per sample (grid.y) first / median / last start and ticket; active workgroups per microsecond
per = (ta.A + 255) // 256
end = st[:, 9]
for b in range(min(B, nwg // per)):
    s_, e_ = (st[b * per:(b + 1) * per, 0] - t0) / 100.0, (end[b * per:(b + 1) * per] - t0) / 100.0
    life = e_ - s_
    print(f"sample {b}: start min/med/max {s_.min():.1f}/{np.median(s_):.1f}/{s_.max():.1f}  ticket med/max {np.median(e_):.1f}/{e_.max():.1f}  "
          f"lifetime med/p90/max {np.median(life):.1f}/{np.percentile(life, 90):.1f}/{life.max():.1f}")
s_all, e_all = (st[:, 0] - t0) / 100.0, (end - t0) / 100.0
print("active workgroups at t us:", " ".join(f"{t_}:{int(((s_all <= t_) & (e_all > t_)).sum())}" for t_ in range(0, 34, 2)))
has_clip = st[:, 3] >= t0
clip = np.where(has_clip, (st[:, 4] - st[:, 3]) / 100.0, 0.0)
for lo, hi in ((0, 0.01), (0.01, 3), (3, 6), (6, 99)):
    m = (clip >= lo) & (clip < hi)
    if m.any():
        print(f"clip time in [{lo},{hi}) us: {m.sum():4d} workgroups, lifetime median {np.median((e_all - s_all)[m]):.1f} us")
