"""k_step's roles timed one at a time (development aid): submit(points) followed by LAG drain calls makes four
launches with ONE stage role each (split / tile / order / emit); their event pairs land in the timing ring in that order."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa: E402,F401
from pp_amd import synth  # noqa: E402
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60000)
ap.add_argument("--half", type=float, default=50.0)
ap.add_argument("--P", type=int, default=12000)
ap.add_argument("--N", type=int, default=100)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--order", type=int, default=1)
ap.add_argument("--step", type=float, default=0.2)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
cfg = VoxelConfig.square(a.half, a.step, a.P, a.N, order=a.order)
vox = PillarVoxelizer(cfg)
pts = torch.from_numpy(np.stack([synth.lidar_like(a.n, a.half, s) for s in range(a.batch)])).cuda()
out = (torch.empty((a.batch, 9, a.P, a.N), dtype=torch.float32, device="cuda"),
       torch.empty((a.batch, a.P, 3), dtype=torch.int64, device="cuda"))
def one():
    vox.submit(pts)
    for _ in range(vox.LAG):
        vox.submit(None, out=out)


for _ in range(5):
    one()
torch.cuda.synchronize()
vox.set_timing(4 * a.iters)
for _ in range(a.iters):
    one()
torch.cuda.synchronize()
ms = np.array(vox.read_emit_ms(4 * a.iters)).reshape(-1, 4) * 1e3
med = np.median(ms, axis=0)
print(f"batch={a.batch} n={a.n} step={a.step} order={a.order}: k_step roles alone (median us): "
      f"split {med[0]:.1f}  tile {med[1]:.1f}  order {med[2]:.1f}  emit {med[3]:.1f}")
