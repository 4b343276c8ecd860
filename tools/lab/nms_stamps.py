"""tools/lab/nms_stamps.py [bias]: per-chunk phase stamps of k_nms (library built with -DPP_NMS_STAMPS), sample 0."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pp_amd import boxes, synth, _lib
from pp_amd.pipeline import PillarPipeline
from pp_amd.postprocess import Detector
from pp_amd.voxelizer import VoxelConfig
cfg = VoxelConfig.square(50.0, 0.2, 12000, 100)
pipe = PillarPipeline(cfg, seed=0)
pipe.model.eval()
with torch.no_grad():
    pipe.model.det_head.cls.bias.fill_(float(sys.argv[1]) if len(sys.argv) > 1 else -1.6)
acfg = pipe.anchor_cfg
det = Detector(boxes.make_anchors(acfg), acfg, 500, 0.2, 0.2, -50.0, -50.0, pos_thresh=0.2, nms_thresh=0.1)
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(1)])).cuda()
cls, reg = pipe.forward_fused(pts)
for _ in range(5):
    out = det(cls[0], reg[0])
torch.cuda.synchronize()
buf = np.zeros(64 * 8, np.uint64)
f = _lib.lib().pp_debug_nms_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data, buf.size) == 0
st = buf.reshape(64, 8).astype(np.int64)
t0 = st[0, 0]
print("kept", int(out[2].item()))
print("chunk: start | keys | kept-test | sync | matrix | sync | greedy | sync   (us, deltas)")
for c in range(64):
    if st[c, 0] < t0 or (c and st[c, 0] <= st[c - 1, 0]):
        break
    d = np.diff(st[c]) / 100.0
    print(f"{c:3d}: {(st[c,0]-t0)/100.0:7.2f} | " + " ".join(f"{x:6.2f}" for x in d))
