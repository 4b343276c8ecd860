"""Committed vectors of the hot path (tests/golden/hotpath_golden.npz, made by make_hotpath_golden.py from
the CPU oracle): the oracle must still reproduce them (CPU), and the HIP path must produce them without
the oracle being involved at all (GPU).  An ORACLE-DRIFT pin, not a reference pin: the vectors are the
oracle's own output (the reference-run vectors are targets_ref_golden.npz and model_golden.npz)."""
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


def test_oracle_reproduces_decode_and_ingest_vectors(oracle, gold):
    from pp_amd import boxes
    a = boxes.make_anchors(boxes.AnchorConfig(16, 16))
    b, k = oracle.postprocess(gold["dec_cls"], gold["dec_reg"], a["centers"], a["wlh"], a["yaw"], a["xy"], 32, 0.2, 0.2,
                              -3.2, -3.2, pos_thresh=0.3, nms_thresh=0.1)
    assert np.array_equal(k.astype(np.int32), gold["dec_kept"]) and len(k) >= 5
    assert np.allclose(b, gold["dec_boxes"], rtol=1e-6, atol=1e-6)         # exp / tanh / asin: libm
    sweeps = [(gold["ing_raw"][s], gold["ing_mats"][s]) for s in range(2)]
    assert np.array_equal(oracle.lidar_ingest(sweeps, min_dist=0.5), gold["ing_points"])


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


@pytest.mark.gpu
def test_hip_decode_and_ingest_produce_the_committed_vectors(gpu, gold):
    import torch
    from pp_amd import boxes
    from pp_amd.ingest import LidarIngest
    from pp_amd.postprocess import Detector
    acfg = boxes.AnchorConfig(16, 16)
    det = Detector(boxes.make_anchors(acfg), acfg, 32, 0.2, 0.2, -3.2, -3.2, pos_thresh=0.3, nms_thresh=0.1, device=gpu)
    b, k, n = det(torch.from_numpy(gold["dec_cls"]).to(gpu), torch.from_numpy(gold["dec_reg"]).to(gpu))
    torch.cuda.synchronize()
    n = int(n.item())
    assert n == len(gold["dec_kept"]) and np.array_equal(k.cpu().numpy()[:n], gold["dec_kept"])
    assert np.allclose(b.cpu().numpy()[:n], gold["dec_boxes"], rtol=1e-5, atol=1e-5)
    out = LidarIngest(device=gpu, min_dist=0.5)([(gold["ing_raw"][s], gold["ing_mats"][s]) for s in range(2)])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    keep = ~np.isnan(got[:, 0])
    ref = gold["ing_points"]
    assert keep.sum() == len(ref) and (~keep).sum() >= 10
    assert np.abs(got[keep].astype(np.float64) - ref).max() <= 2e-6 * max(1.0, np.abs(ref[:, :3]).max())
    assert np.array_equal(got[keep][:, 3], ref[:, 3].astype(np.float32))
