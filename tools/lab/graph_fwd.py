"""tools/lab/graph_fwd.py: does a hipGraph of the NETWORK part of the forward (PillarPipeline.model on the static
output buffers of forward_pipelined) shorten the end-to-end step?  Eager loop against submit + graph replay; outputs
compared bit for bit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd  # noqa
from pp_amd import synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
pipe = PillarPipeline(VoxelConfig.square(50.0, 0.2, 12000, 100), device=dev, seed=0)
pipe.model.eval()
sets = [torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, 1000 * r + s) for s in range(B)])).to(dev) for r in range(4)]


def eager(steps):
    for i in range(steps):
        out = pipe.forward_pipelined(sets[i % 4])
    return out


for _ in range(3):
    eager(12)
torch.cuda.synchronize()
t0 = time.perf_counter(); ref = eager(40); torch.cuda.synchronize(); dt_e = (time.perf_counter() - t0) / 40
ref = tuple(x.clone() for x in ref)

# graph of the network on the static buffers
bufs = pipe._buffers(B)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(3):
        pipe.model(bufs[0], bufs[1])
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    gout = pipe.model(bufs[0], bufs[1])
torch.cuda.synchronize()


def graphed(steps):
    for i in range(steps):
        r = pipe.voxelizer.submit(sets[i % 4], out=bufs)
        if r is not None:
            g.replay()
    return gout


pipe.voxelizer.reset_stream()
for _ in range(2):
    graphed(12)
torch.cuda.synchronize()
pipe.voxelizer.reset_stream()
graphed(4)   # refill so that step 40 of the timed loop lines up with the eager loop's last batch
torch.cuda.synchronize()
pipe.voxelizer.reset_stream()
t0 = time.perf_counter(); out = graphed(40 + 0); torch.cuda.synchronize(); dt_g = (time.perf_counter() - t0) / 40
print(f"B={B}: eager {dt_e*1e3:.3f} ms/step ({B/dt_e:.1f} sweeps/s)   network as a hipGraph {dt_g*1e3:.3f} ms/step ({B/dt_g:.1f} sweeps/s)")
# same batch through both: run one more matched pair
pipe.voxelizer.reset_stream()
for i in range(4):
    e = pipe.forward_pipelined(sets[i % 4])
e = tuple(x.clone() for x in e)
pipe.voxelizer.reset_stream()
for i in range(4):
    r = pipe.voxelizer.submit(sets[i % 4], out=bufs)
    if r is not None:
        g.replay()
torch.cuda.synchronize()
print("outputs equal:", all(torch.equal(a, b) for a, b in zip(e, gout)),
      "max |diff|:", [float((a - b).abs().max()) for a, b in zip(e, gout)], "max |value|:", [float(a.abs().max()) for a in e])
# the graph against itself (replayed twice on the same input) and eager against itself
o1 = tuple(x.clone() for x in gout); g.replay(); torch.cuda.synchronize()
print("graph replay repeatable:", all(torch.equal(a, b) for a, b in zip(o1, gout)))
e2 = pipe.model(bufs[0], bufs[1]); torch.cuda.synchronize()
print("eager on the same buffers == graph:", all(torch.equal(a, b) for a, b in zip(e2, gout)),
      [float((a - b).abs().max()) for a, b in zip(e2, gout)])
