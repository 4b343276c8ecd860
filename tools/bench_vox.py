"""Voxelizer-only micro benchmark (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa: E402
from pp_amd import synth  # noqa: E402
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60000)
ap.add_argument("--half", type=float, default=50.0)
ap.add_argument("--P", type=int, default=12000)
ap.add_argument("--N", type=int, default=100)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--order", type=int, default=1)
ap.add_argument("--step", type=float, default=0.2)
a = ap.parse_args()

cfg = VoxelConfig.square(a.half, a.step, a.P, a.N, order=a.order)
vox = PillarVoxelizer(cfg)
pts = torch.from_numpy(np.stack([synth.lidar_like(a.n, a.half, s) for s in range(a.batch)])).cuda()
out = (torch.empty((a.batch, 9, a.P, a.N), dtype=torch.float32, device="cuda"),
       torch.empty((a.batch, a.P, 3), dtype=torch.int64, device="cuda"))
for _ in range(20):
    vox(pts, out=out)
torch.cuda.synchronize()
vox.set_timing(a.iters)
t0 = time.perf_counter()
for _ in range(a.iters):
    vox(pts, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
ms = vox.read_emit_ms(a.iters)
byt = cfg.algorithmic_bytes(a.n) * a.batch
emit = float(np.median(ms)) * 1e-3
print(f"batch={a.batch} n={a.n} P={a.P} N={a.N}: {dt*1e6:.1f} us/step  "
      f"{a.batch/dt:.0f} sweeps/s  emit median {emit*1e6:.1f} us "
      f"-> {byt/emit/1e12:.2f} TB/s algorithmic ({byt/emit/8e12*100:.1f}% of 8 TB/s); "
      f"whole pipeline {byt/dt/1e12:.2f} TB/s")
