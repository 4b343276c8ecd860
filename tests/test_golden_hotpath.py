"""Committed vectors of the hot path (tests/golden/hotpath_golden.npz, made by make_hotpath_golden.py from
the CPU oracle): the oracle must still reproduce them (CPU), and the HIP path must produce them without
the oracle being involved at all (GPU)."""
import os

import numpy as np
import pytest

from util import grid_args

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def test_oracle_reproduces_the_committed_vectors(oracle, gold):
    half, step, P, N = gold["vox_geom"]
    pts = gold["vox_points"]
    for order in (0, 1):
        for s in range(len(pts)):
            p, i, _ = oracle.dataset_voxel_stage(pts[s].astype(np.float64), int(P), int(N),
                                                 *grid_args(float(half), float(step)), order=order)
            assert np.array_equal(p, gold[f"vox_pillars_o{order}"][s])
            assert np.array_equal(i, gold[f"vox_indices_o{order}"][s])
    cls_t, reg_t, ious = oracle.create_target(
        gold["tgt_anchor_corners"], gold["tgt_gt_corners_img"], gold["tgt_anchor_centers"], gold["tgt_gt_centers_img"],
        *_anchor_wlh_yaw(gold), gold["tgt_gt_centers"], gold["tgt_gt_wlh"], gold["tgt_gt_yaw"],
        gold["tgt_gt_classes"], int(gold["tgt_geom"][2]), pos_thresh=0.5)
    assert np.array_equal(ious, gold["tgt_ious"])                  # no libm in the clipper: exact
    assert np.array_equal(cls_t, gold["tgt_cls"])
    assert np.allclose(reg_t, gold["tgt_reg"], rtol=0, atol=1e-12)  # log / sin: libm


def _anchor_wlh_yaw(gold):
    from pp_amd import boxes
    a = boxes.make_anchors(boxes.AnchorConfig(int(gold["tgt_geom"][0]), int(gold["tgt_geom"][1])))
    assert np.array_equal(a["corners"], gold["tgt_anchor_corners"])
    return a["wlh"], a["yaw"]


@pytest.mark.gpu
def test_hip_path_produces_the_committed_vectors(gpu, gold):
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    half, step, P, N = gold["vox_geom"]
    pts = torch.from_numpy(gold["vox_points"]).to(gpu)
    for order in (0, 1):
        vox = PillarVoxelizer(VoxelConfig.square(float(half), float(step), int(P), int(N), order=order), device=gpu)
        pil, idx = vox(pts)
        torch.cuda.synchronize()
        assert np.array_equal(pil.cpu().numpy(), gold[f"vox_pillars_o{order}"])
        assert np.array_equal(idx.cpu().numpy(), gold[f"vox_indices_o{order}"])
    acfg = boxes.AnchorConfig(int(gold["tgt_geom"][0]), int(gold["tgt_geom"][1]))
    H = int(gold["tgt_geom"][2])
    for src in (acfg, boxes.make_anchors(acfg)):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=0.5, device=gpu)
        cls_t, reg_t = ta.assign(gold["tgt_gt_centers"], gold["tgt_gt_wlh"], gold["tgt_gt_yaw"], gold["tgt_gt_classes"],
                                 check=True)
        torch.cuda.synchronize()
        assert np.array_equal(cls_t.cpu().numpy(), gold["tgt_cls"].astype(np.float32))
        assert np.abs(reg_t.cpu().numpy() - gold["tgt_reg"].astype(np.float32)).max() <= 1e-6
        if not isinstance(src, boxes.AnchorConfig):
            d = ta.ious(gold["tgt_gt_corners_img"], gold["tgt_gt_centers_img"]).cpu().numpy()
            assert np.array_equal(d, gold["tgt_ious"])
