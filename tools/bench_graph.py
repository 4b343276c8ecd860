"""Is the voxelizer (and the target assignment) capturable in a HIP graph, and what does a replay cost?
Captures one call of each on a side stream with torch.cuda.graph, replays, compares with the direct call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import boxes, synth
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
from pp_amd.targets import TargetAssigner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = VoxelConfig.square(50.0, 0.2, 12000, 100)
vox = PillarVoxelizer(cfg)
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
for _ in range(5):
    ref = vox(pts)
torch.cuda.synchronize()
def wall(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
direct = wall(lambda: vox(pts))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    vox(pts)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = vox(pts)
torch.cuda.synchronize()
for t in out: t.zero_()
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(out, ref))
rep = wall(g.replay)
print(f"voxelizer B={B}: direct {direct:.1f} us per call, graph replay {rep:.1f} us, outputs equal: {same}")
acfg = boxes.AnchorConfig(250, 250)
ta = TargetAssigner(acfg, canvas_height=500)
gt = synth.gt_boxes(40, 500, 0)
gg = ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
for _ in range(5):
    ref_t = ta.assign_device(*gg)
torch.cuda.synchronize()
direct_t = wall(lambda: ta.assign_device(*gg))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    ta.assign_device(*gg)
    torch.cuda.synchronize()
    with torch.cuda.graph(g2, stream=s):
        out_t = ta.assign_device(*gg)
torch.cuda.synchronize()
for t in out_t: t.fill_(7)
g2.replay()
torch.cuda.synchronize()
same_t = all(torch.equal(a, b) for a, b in zip(out_t, ref_t))
rep_t = wall(g2.replay)
print(f"target assignment: direct {direct_t:.1f} us per call, graph replay {rep_t:.1f} us, outputs equal: {same_t}")
