import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import pp_amd
from pp_amd import boxes, synth, _lib
from pp_amd.targets import TargetAssigner
cfg = boxes.AnchorConfig(250, 250)
ta = TargetAssigner(cfg, canvas_height=500)
counts, packed = ta.upload_batch([synth.gt_boxes(40, 500, 0)])
for _ in range(20):
    ta.assign_batch_device(counts, packed)
torch.cuda.synchronize()
buf = np.zeros(16 * 4096, np.uint64)
f = _lib.lib().pp_debug_iou_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 16)[:489].astype(np.int64)
t0 = st[:, 0].min()
pos = st[:, 14] >= st[:, 5]   # WGs that took the staged path in THIS launch
pos &= st[:, 14] <= st[:, 6]
print("workgroups with positives:", pos.sum())
names = {5: "reduced", 11: "cls staged", 12: "cls stored+barrier", 13: "reg zeroed", 14: "math done", 6: "rows out"}
seq = [5, 11, 12, 13, 14, 6]
for a, b in zip(seq[:-1], seq[1:]):
    d = (st[pos, b] - st[pos, a]) / 100.0
    print(f"{names[a]:>20s} -> {names[b]:20s} median {np.median(d):5.2f} max {d.max():5.2f} us")
d = (st[pos, 6] - st[pos, 5]) / 100.0
print("rows phase total: median %.2f max %.2f" % (np.median(d), d.max()))
