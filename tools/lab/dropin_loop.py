"""tools/lab/dropin_loop.py [calls]: the pybind11 module's create_pillars / make_ious on host arrays in a loop, for
rocprofv3 --kernel-trace --memory-copy-trace --stats (which launches and copies a call consists of)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pp_amd  # noqa: E402
from pp_amd import _lib, boxes, synth  # noqa: E402

spec = importlib.util.spec_from_file_location("pillars", _lib.pybind_module_path())
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 50
what = sys.argv[2] if len(sys.argv) > 2 else "both"
pts = synth.lidar_like(60000, 50.0, 0).astype(np.float64)
agg = np.ascontiguousarray(pts.T)
P, N = 12000, 100
args = (N, P, .2, .2, -50., -50., -10., 50., 50., 10., 500)
T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
if what in ("both", "pillars"):
    for _ in range(calls):
        mod.create_pillars(agg.transpose([1, 0]), T, I, *args)
if what in ("both", "ious"):
    anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
    gt = synth.gt_boxes(40, 500, 0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)
    ious = np.zeros((125000, 40))
    for _ in range(calls):
        mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious)
print("done")
