"""tools/lab/glue_thp.py: the reference caller's own statements around the two module calls (data/dataset.py:88-106,
utils/box_utils.py:180-183) with FRESH np.zeros outputs per sample -- the first touch of those arrays is inside the call --
and how much of the statements' time the call is.  (Round 6 used it to try madvise(MADV_HUGEPAGE) on the caller's arrays,
knob PP_DROPIN_THP: no effect, not kept -- profiles/r06/NOTES.md section 3.)"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pp_amd  # noqa: E402
from pp_amd import _lib, boxes, synth  # noqa: E402

spec = importlib.util.spec_from_file_location("pillars", _lib.pybind_module_path())
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
pts = synth.lidar_like(60000, 50.0, 0).astype(np.float64)
agg = np.ascontiguousarray(pts.T)
P, N = 12000, 100
args = (N, P, .2, .2, -50., -50., -10., 50., 50., 10., 500)
anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
gt = synth.gt_boxes(40, 500, 0)
c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)


def glue():
    pillar = np.zeros((P, N, 9))
    indices = np.zeros((P, 3))
    t0 = time.perf_counter()
    mod.create_pillars(agg.transpose([1, 0]), pillar, indices, *args)
    t1 = time.perf_counter()
    pillar = torch.from_numpy(pillar.transpose([2, 0, 1])).float()
    indices = torch.from_numpy(indices).long()
    return t1 - t0, pillar


def ious_glue():
    ious = np.zeros((anchors["corners"].shape[0], 40))
    t0 = time.perf_counter()
    mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious)
    return time.perf_counter() - t0, ious


for fn, name in ((glue, "dataset.py:88-106 statements"), (ious_glue, "np.zeros + make_ious")):
    whole, call = [], []
    for rep in range(12):
        t0 = time.perf_counter()
        c, keep = fn()
        whole.append(time.perf_counter() - t0)
        call.append(c)
        del keep
    print(f"THP={os.environ.get('PP_DROPIN_THP', '1')} {name}: median {np.median(whole[2:]) * 1e3:.2f} ms, of it the call "
          f"{np.median(call[2:]) * 1e3:.2f} ms", flush=True)
