"""tools/lab/iou_stamps.py: phase stamps of one k_targets launch (library built with -DPP_IOU_STAMPS)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd
from pp_amd import boxes, synth, _lib
from pp_amd.targets import TargetAssigner
fm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = boxes.AnchorConfig(fm, fm)
gt = synth.gt_boxes(G, 2 * fm, 0)
ta = TargetAssigner(cfg, canvas_height=2 * fm)
g = ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
for _ in range(20):
    ta.assign_device(*g)
torch.cuda.synchronize()
nwg = (ta.A + 255) // 256
buf = np.zeros(16 * 4096, np.uint64)
f = _lib.lib().pp_debug_iou_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 16)[:nwg].astype(np.int64)
t0 = st[:, 0].min()
names = ["start", "staged", "gated", "queued", "clipped", "reduced", "rows out", "appended", "drained", "ticket", "tail", "t:loaded", "t:argmax", "t:forced?", "t:rows0", "t:rows"]
print(f"{nwg} workgroups; 10 ns ticks relative to the first start; per stamp: min / median / max over the workgroups that have it")
for k, nm in enumerate(names):
    col = st[:, k]
    ok = col >= t0
    if ok.any():
        c = (col[ok] - t0) / 100.0
        print(f"  {nm:9s} n={ok.sum():4d}  {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f} us")
dur = (st[:, 9] - st[:, 0]) / 100.0
print("workgroup lifetime: median %.2f max %.2f us; clip phase (queued->clipped) median %.2f max %.2f us" % (
    np.median(dur), dur.max(), np.median((st[:, 4] - st[:, 3])[st[:, 3] >= t0]) / 100.0, ((st[:, 4] - st[:, 3])[st[:, 3] >= t0]).max() / 100.0))
