"""Dev tool: time pp_pfn_dense_dev alone on a config-2 dense tensor (and a plain read of it)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import synth, _lib
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
import pp_amd.model as M
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vox = PillarVoxelizer(VoxelConfig.square(50.0, 0.2, 12000, 100))
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
dense, idx = vox(pts)
fn = M.PPFeatureNet(9, 64).cuda().eval()
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3
with torch.no_grad():
    us = t(lambda: fn(dense))
    print(f"[{os.environ.get('PP_HIP_LIB', 'default')}] pfn_dense B={B}: {us:.1f} us  ({dense.numel()*4/us/1e6:.2f} TB/s read, "
          f"{2*9*64*dense.numel()/9/us/1e6:.1f} TFLOP/s)")
    us = t(lambda: dense.amax())
    print(f"amax over the dense tensor: {us:.1f} us ({dense.numel()*4/us/1e6:.2f} TB/s)")
