"""Dev tool: single-sweep inference latency, eager vs captured in a HIP graph."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pp_amd import synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.backends.cudnn.benchmark = True
pipe = PillarPipeline(VoxelConfig.square(50.0, 0.2, 12000, 100), seed=0)
pipe.model.eval()
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
def lat(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for name, fwd in (("fused", pipe.forward_fused), ("dense", pipe.forward)):
    e = lat(lambda: fwd(pts))
    static = pts.clone()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fwd(static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd(static)
    gl = lat(lambda: g.replay())
    ref = fwd(static)
    g.replay(); torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"B={B} {name}: eager {e:.3f} ms, graph {gl:.3f} ms per step (sync to sync), graph == eager: {ok}")
