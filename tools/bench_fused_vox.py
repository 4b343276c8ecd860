"""Fused feature-net voxelizer call alone (development aid): kernel durations of pfn / pfn_canvas (reuse)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa
from pp_amd import synth, _lib
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pipe = PillarPipeline(VoxelConfig.square(50.0, 0.2, 12000, 100), seed=0)
pipe.model.eval()
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
tab = pipe.model.feature_net.fused_table(pipe.device)
H, W = pipe.model.scatter.h, pipe.model.scatter.w
out = (pipe._canvas(B, H, W), torch.empty((B, 12000, 3), dtype=torch.int64, device="cuda"))
vox = pipe.voxelizer
for reuse in (False, True):
    for _ in range(10):
        vox.pfn_canvas(pts, tab, (H, W), out=out, reuse=reuse)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        vox.pfn_canvas(pts, tab, (H, W), out=out, reuse=reuse)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    vox.set_timing(64)
    for _ in range(64):
        vox.pfn_canvas(pts, tab, (H, W), out=out, reuse=reuse)
    torch.cuda.synchronize()
    ks = np.mean(vox.read_kernel_ms(_lib.KERNEL_SPLIT)) * 1e3
    kt = np.mean(vox.read_kernel_ms(_lib.KERNEL_TILE)) * 1e3
    ke = np.mean(vox.read_kernel_ms(_lib.KERNEL_EMIT)) * 1e3
    vox.set_timing(0)
    print(f"fused voxelizer B={B} reuse={reuse}: {dt*1e6:.1f} us per call; k_split {ks:.1f} k_tile {kt:.1f} k_emit<pfn> {ke:.1f}")
# the software-pipelined form: ONE launch per call (k_step<pfn>: split | tile | order | fused emit | clear)
for _ in range(10):
    vox.submit_pfn_canvas(pts, tab, (H, W))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    vox.submit_pfn_canvas(pts, tab, (H, W))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
vox.set_timing(64)
for _ in range(64):
    vox.submit_pfn_canvas(pts, tab, (H, W))
torch.cuda.synchronize()
ke = np.mean(vox.read_kernel_ms(_lib.KERNEL_EMIT)) * 1e3
vox.set_timing(0)
vox.reset_stream()
byt = (16 * 60000 + 2 * (256 * 12000 + 24 * 12000)) * B
print(f"fused voxelizer B={B} pipelined (k_step<pfn>): {dt*1e6:.1f} us per call; kernel {ke:.1f} us; its bytes {byt/1e6:.1f} MB "
      f"-> {byt/dt/8e12:.3f} of 8 TB/s")
