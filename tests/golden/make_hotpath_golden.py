"""Generates tests/golden/hotpath_golden.npz: small seeded inputs of the hot path with the outputs of the
CPU oracle (oracle/, the restatement of data/pillars.cpp and utils/box_utils.py) at the time of writing.

The reference ships no fixtures for this path and cannot be built in this image (Boost headers absent,
DESIGN.md), so these vectors are an ORACLE-DRIFT PIN ONLY: they hold the oracle (and the HIP path, in a
comparison that does not run the oracle) to what the oracle produced when they were written.  They do NOT pin
parity with the reference's pillars.cpp / box_utils.py -- what does, as far as this image allows:
tests/golden/targets_ref_golden.npz (create_target / make_target run from the reference's own source),
tests/golden/model_golden.npz (model.py / loss.py run from the reference's own source), and the
hand-computed known answers of tests/test_oracle_*.py and tests/test_host_logic.py (SURVEY 5.9 / 8c probe
values, analytic IoUs, hand-derived corner / anchor rows).  They are data: inputs and expected outputs.

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
# inference post-processing: a 16x16x2 head output, threshold 0.3 (evaluate.py:231-245)
rng = np.random.default_rng(77)
dcfg = boxes.AnchorConfig(16, 16)
danch = boxes.make_anchors(dcfg)
dcls = rng.normal(-1.6, 1.5, (dcfg.per_cell * 9, 16, 16)).astype(np.float32)
dreg = rng.normal(0.0, 0.3, (dcfg.per_cell * 8, 16, 16)).astype(np.float32)
dboxes, dkept = O.postprocess(dcls, dreg, danch["centers"], danch["wlh"], danch["yaw"], danch["xy"], 32, 0.2, 0.2,
                              -3.2, -3.2, pos_thresh=0.3, nms_thresh=0.1)
out.update(dec_cls=dcls, dec_reg=dreg, dec_boxes=dboxes, dec_kept=dkept.astype(np.int32))

# lidar ingest: two sweeps of 5-column rows, rigid transforms, remove_close (dataset.py:51-88)
def _tm(t, yaw):
    m = np.eye(4)
    m[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
    m[:3, 3] = t
    return m
sweeps = []
for sidx in range(2):
    raw = np.zeros((400, 5), np.float32)
    raw[:, :4] = synth.lidar_like(400, 8.0, 300 + sidx)
    raw[:, 4] = rng.integers(0, 64, 400)
    mat = _tm([0.4 * sidx, -0.2, 0.1], 0.03 * (sidx + 1))
    raw[10:20, :3] = (np.linalg.inv(mat) @ np.column_stack([rng.uniform(-0.4, 0.4, (10, 2)), np.zeros(10), np.ones(10)]).T).T[:, :3]
    sweeps.append((raw, mat))
ing = O.lidar_ingest(sweeps, min_dist=0.5)
out.update(ing_raw=np.stack([r for r, _ in sweeps]), ing_mats=np.stack([m for _, m in sweeps]), ing_points=ing)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes;",
      {k: v.shape for k, v in out.items() if k.startswith(("vox_pillars", "tgt_ious", "tgt_reg"))},
      "positives", int((reg_t[:, 0] == 1).sum()), "nonzero ious", int((ious > 0).sum()),
      "decoded", len(dkept), "ingested", ing.shape)
