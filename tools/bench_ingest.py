"""Lidar ingest micro benchmark (pp_ingest_dev): one 60 000-point sweep with 5 columns per raw row."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd.ingest import LidarIngest, transform_matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(0)
raw = [torch.from_numpy(rng.normal(0, 30, (n, 5)).astype(np.float32)).cuda() for _ in range(ns)]
th = 0.3
mat = transform_matrix([1.0, 2.0, 0.5], [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
ing = LidarIngest()
sweeps = [(r, mat) for r in raw]
for _ in range(10):
    out = ing(sweeps)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    out = ing(sweeps)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 5
nb = ns * n * (20 + 16)
print(f"ingest {ns} x {n} points: {us:.1f} us per sample ({us/ns:.1f} per sweep); {nb/1e6:.2f} MB -> {nb/us/1e6:.2f} TB/s")
