"""Headline-path forward loop (development aid for profiling): voxelizer -> dense tensor ->
PPFeatureNet -> scatter -> backbone -> head."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pp_amd import synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.backends.cudnn.benchmark = True
torch.backends.cudnn.deterministic = bool(int(os.environ.get("PP_DETERMINISTIC", "0")))
pipe = PillarPipeline(VoxelConfig.square(50.0, 0.2, 12000, 100), seed=0)
pipe.model.eval()
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
for _ in range(8):
    pipe.forward(pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    pipe.forward(pts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"headline forward B={B}: {dt*1e3:.2f} ms/step, {B/dt:.0f} sweeps/s")
