"""Target-assignment micro benchmark at the reference's default anchor set (300x300x6 = 540 000 anchors)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import boxes, synth
from pp_amd.targets import TargetAssigner
G = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = boxes.AnchorConfig.reference_default()
gt = synth.gt_boxes(G, 600, 0)
ta = TargetAssigner(cfg, canvas_height=600)
g = ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
for _ in range(10):
    ta.assign_device(*g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = 200
e0.record()
for _ in range(it):
    ta.assign_device(*g)
e1.record()
torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / it
print(f"A={ta.A} G={G}: {dt*1e6:.1f} us per sample; algorithmic 112*A = {112*ta.A/1e6:.1f} MB -> {112*ta.A/dt/1e9:.0f} GB/s")
