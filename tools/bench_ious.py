"""Dense IoU matrix micro benchmark (pp_make_ious_dev): BASELINE config 3, A=125 000, G=40 -> [A,G] f64 = 40 MB."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import boxes, synth
from pp_amd.targets import TargetAssigner
fm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = boxes.AnchorConfig(fm, fm)
ta = TargetAssigner(boxes.make_anchors(cfg), canvas_height=2 * fm)
gt = synth.gt_boxes(G, 2 * fm, 0)
c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 2 * fm)
for _ in range(5):
    out = ta.ious(k_img, c_img, check=False)
torch.cuda.synchronize()
f64 = dict(dtype=torch.float64, device=ta.device)
import ctypes
from pp_amd import _lib
from pp_amd.targets import _vp
gc = torch.as_tensor(np.ascontiguousarray(k_img), **f64).contiguous()
gn = torch.as_tensor(np.ascontiguousarray(c_img), **f64).contiguous()
out = torch.empty((ta.A, G), **f64)
stream = ctypes.c_void_p(torch.cuda.current_stream(ta.device).cuda_stream)
def call():
    _lib.lib().pp_make_ious_dev(ta._ctx.handle, stream, _vp(ta.a_corners), _vp(ta.a_centers), 3, ta.A, _vp(gc), _vp(gn), 3, G, _vp(out))
for _ in range(10):
    call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 10
print(f"pp_make_ious_dev A={ta.A} G={G}: {us:.1f} us per call; {ta.A*G*8/1e6:.0f} MB out -> {ta.A*G*8/us/1e6:.2f} TB/s; nonzero {int((out>0).sum())}")
