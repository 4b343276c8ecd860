"""Generates tests/golden/hotpath_golden.npz: small seeded inputs of the hot path with the outputs of the
CPU oracle (oracle/, the restatement of data/pillars.cpp and utils/box_utils.py) at the time of writing.

The reference ships no fixtures for this path and cannot be built in this image (Boost headers absent,
DESIGN.md), so these vectors pin the ORACLE -- against drift between rounds -- and give the GPU tests a
comparison that does not run the oracle at all.  They are data: inputs and expected outputs.

Run:  python tests/golden/make_hotpath_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pp_amd  # noqa: E402,F401
from oracle import oracle as O  # noqa: E402
from pp_amd import boxes, synth  # noqa: E402
from util import grid_args  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.npz")
O.build()
out = {}

# voxelizer: 2 sweeps x 3000 points, 32x32 grid (more occupied cells than P: the overflow rule), both orders
HALF, STEP, P, N = 8.0, 0.5, 200, 6
pts = np.stack([synth.lidar_like(3000, HALF, 100 + s) for s in range(2)]).astype(np.float32)
pts[0, :40, :2] = pts[0, 0, :2]                       # a crowded cell: more points than N
pts[1, 5] = [HALF, 0.0, 0.0, 1.0]                     # x == x_max: outside the half-open range
pts[1, 6] = [-HALF, -HALF, -10.0, 1.0]                # the lowest corner: inside
out.update(vox_points=pts, vox_geom=np.array([HALF, STEP, P, N]))
for order in (0, 1):
    pil, idx = [], []
    for s in range(2):
        p, i, _ = O.dataset_voxel_stage(pts[s].astype(np.float64), P, N, *grid_args(HALF, STEP), order=order)
        pil.append(p)
        idx.append(i)
    out[f"vox_pillars_o{order}"] = np.stack(pil)
    out[f"vox_indices_o{order}"] = np.stack(idx)

# IoU + targets: 12x12 feature map x 2 anchors, 7 boxes (one sits exactly on an anchor, two are duplicates)
acfg = boxes.AnchorConfig(12, 12)
anchors = boxes.make_anchors(acfg)
H = 24
gt = synth.gt_boxes(7, H, 5, margin=4.0)
gt["centers"][1] = anchors["centers"][57]
gt["wlh"][1] = anchors["wlh"][57]
gt["yaw"][1] = anchors["yaw"][57]
for k in ("centers", "wlh", "yaw"):
    gt[k][3] = gt[k][2]
c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
cls_t, reg_t, ious = O.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                     anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                                     pos_thresh=0.5)
out.update(tgt_gt_centers=gt["centers"], tgt_gt_wlh=gt["wlh"], tgt_gt_yaw=gt["yaw"],
           tgt_gt_classes=gt["classes"].astype(np.int32), tgt_gt_corners_img=k_img, tgt_gt_centers_img=c_img,
           tgt_anchor_corners=anchors["corners"], tgt_anchor_centers=anchors["centers"],
           tgt_ious=ious, tgt_cls=cls_t, tgt_reg=reg_t, tgt_geom=np.array([12, 12, H, 0.5]))
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes;",
      {k: v.shape for k, v in out.items() if k.startswith(("vox_pillars", "tgt_ious", "tgt_reg"))},
      "positives", int((reg_t[:, 0] == 1).sum()), "nonzero ious", int((ious > 0).sum()))
