"""Dev tool: post-processing (pp_decode_strided_dev) time per sample on config-2 network outputs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pp_amd import boxes, synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.postprocess import Detector
from pp_amd.voxelizer import VoxelConfig
torch.backends.cudnn.benchmark = True
cfg = VoxelConfig.square(50.0, 0.2, 12000, 100)
pipe = PillarPipeline(cfg, seed=0)
pipe.model.eval()
with torch.no_grad():
    pipe.model.det_head.cls.bias.fill_(float(sys.argv[1]) if len(sys.argv) > 1 else -2.0)
acfg = pipe.anchor_cfg
det = Detector(boxes.make_anchors(acfg), acfg, 500, 0.2, 0.2, -50.0, -50.0, pos_thresh=0.2, nms_thresh=0.1)
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(4)])).cuda()
cls, reg = pipe.forward_fused(pts)
for _ in range(5):
    out = [det(cls[i], reg[i]) for i in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    out = [det(cls[i], reg[i]) for i in range(4)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
ncand = [int((torch.sigmoid(cls[i].float()).reshape(acfg.per_cell, 9, -1).amax(1) > 0.2).sum().item()) for i in range(4)]
print(f"decode: {dt*1e6:.1f} us per sample; candidates {ncand}, kept {[int(o[2].item()) for o in out]}")
for _ in range(5):
    outb = det(cls, reg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    outb = det(cls, reg)
torch.cuda.synchronize()
dtb = (time.perf_counter() - t0) / 50
same = all(torch.equal(outb[1][i], out[i][1]) for i in range(4))
print(f"decode, batch of 4 in one call: {dtb*1e6:.1f} us per batch ({dtb*1e6/4:.1f} per sample); equal to the per-sample calls: {same}")
