"""Dev tool: training-mode step loop (voxelizer + targets + forward + loss + backward), for profiling.
usage: bench_train.py [B] [channels_last 0/1] [cudnn.benchmark 0/1]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pp_amd import synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CL = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.backends.cudnn.benchmark = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
cfg = VoxelConfig.square(50.0, 0.2, 12000, 100)
pipe = PillarPipeline(cfg, seed=0, with_targets=True)
pipe.model.train()
if CL:
    pipe.model.to(memory_format=torch.channels_last)
    pipe.train_channels_last = True
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
gts = [pipe.upload_ground_truth(synth.gt_boxes(40, cfg.canvas_height, s)) for s in range(B)]
def step():
    pipe.model.zero_grad(set_to_none=True)
    return pipe.train_forward_backward(pts, gts)
t0 = time.perf_counter()
for i in range(4):
    step()
    torch.cuda.synchronize()
    print(f"warm {i}: {time.perf_counter() - t0:.1f}s", flush=True)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"train step B={B} channels_last={CL} benchmark={torch.backends.cudnn.benchmark}: {dt*1e3:.2f} ms/step, {B/dt:.1f} sweeps/s")
