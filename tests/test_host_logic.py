"""Host-side logic that needs no GPU: configs, anchors, synthetic data, sharding."""
import numpy as np


def test_voxel_config_and_algorithmic_bytes():
    from pp_amd.voxelizer import VoxelConfig
    c2 = VoxelConfig.square(50.0, 0.2, 12000, 100)
    assert c2.canvas_height == 500 and c2.canvas_width == 500
    assert c2.algorithmic_bytes(60000) == 44_448_000            # SURVEY 8(d), BASELINE.md section 3
    c5 = VoxelConfig.square(100.0, 0.2, 30000, 100)
    assert c5.algorithmic_bytes(200000) == 111_920_000
    d = VoxelConfig.reference_default()                          # config.py:46-53,60,119-120
    assert (d.max_points_per_pillar, d.max_pillars, d.canvas_height, d.canvas_width) == (200, 24000, 600, 600)


def test_anchor_constants_match_config():
    from pp_amd import boxes
    # values the survey probed from config.py:64-116
    assert np.allclose(boxes.SMALL, [3.75, 7.8125, 1.3125])
    assert np.allclose(boxes.MED, [10, 25, 1.75]) and np.allclose(boxes.LARGE, [14.0625, 46.875, 3.25])
    ref = boxes.AnchorConfig.reference_default()
    assert ref.num_anchors == 540000 and ref.per_cell == 6
    c3 = boxes.AnchorConfig(250, 250)
    assert c3.num_anchors == 125000                              # BASELINE config 3


def test_anchor_geometry_hand_computed():
    """boxes.py against values worked out by hand (not against the oracle): one 0-degree and one
    90-degree MED anchor of feature-map cell (x=0, y=0), fm_scale 0.5 -> centre (1, 1), w=10,
    l=25 (utils/box_utils.py:136-150; lyft Box.bottom_corners order, recalled)."""
    from pp_amd import boxes
    a = boxes.make_anchors(boxes.AnchorConfig(3, 4))
    assert a["corners"].shape == (24, 4, 2) and a["centers"].shape == (24, 3)
    assert a["centers"][0].tolist() == [1.0, 1.0, 0.75] and a["centers"][1].tolist() == [1.0, 1.0, 0.75]
    # index = (y*fm_width + x)*per_cell + d  (box_utils.py:133-135; loss.py:31-36 permute order)
    assert a["centers"][(2 * 4 + 3) * 2 + 1].tolist() == [7.0, 5.0, 0.75]
    # yaw 0: (+l/2,-w/2), (+l/2,+w/2), (-l/2,+w/2), (-l/2,-w/2) about (1,1)
    assert a["corners"][0].tolist() == [[13.5, -4.0], [13.5, 6.0], [-11.5, 6.0], [-11.5, -4.0]]
    # yaw 90: the same template turned a quarter: (x,y) -> (-y,x)
    assert np.allclose(a["corners"][1], [[6.0, 13.5], [-4.0, 13.5], [-4.0, -11.5], [6.0, -11.5]], atol=1e-12)
    assert a["yaw"][0] == 0.0 and a["yaw"][1] == np.pi / 2 and a["wlh"][1].tolist() == [10.0, 25.0, 1.75]

    def signed_area(q):
        x, y = q[:, 0], q[:, 1]
        return 0.5 * (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    # counter-clockwise, positive area w*l: what Polygon_cc of pillars.cpp:16,149-152 expects
    for q in a["corners"]:
        assert abs(signed_area(q) - 250.0) < 1e-9
    # anchor_xy rows (box_utils.py:152-155): corners [2],[0] (yaw 0) / [1],[3] (yaw > 0) are the
    # top-left and bottom-right corner: x1 < x2 always, and y1 < y2 once box_nms has flipped y
    # (evaluate.py:134-135).  The reference's index choice only gives valid rectangles for THIS
    # corner order -- the one pin of the recalled lyft layout that the reference itself holds.
    assert a["xy"][0].tolist() == [-11.5, 6.0, 13.5, -4.0]
    assert np.allclose(a["xy"][1], [-4.0, 13.5, 6.0, -11.5], atol=1e-12)
    H = 8
    xy = a["xy"].copy()
    xy[:, 1], xy[:, 3] = (H - 1) - xy[:, 1], (H - 1) - xy[:, 3]
    assert (xy[:, 0] < xy[:, 2]).all() and (xy[:, 1] < xy[:, 3]).all()
    assert np.allclose((xy[:, 2] - xy[:, 0]) * (xy[:, 3] - xy[:, 1]), 250.0)


def test_ground_truth_image_space_hand_computed():
    """boxes_to_image_space (utils/box_utils.py:19-32): y -> (H-1) - y for centres and corners,
    which turns the counter-clockwise ground-truth quad clockwise (Polygon of pillars.cpp:15)."""
    from pp_amd import boxes
    c, k = boxes.boxes_to_image_space([[3.0, 4.0, 1.0]], [[2.0, 6.0, 1.0]], [0.0], 30)
    assert c.tolist() == [[3.0, 25.0, 1.0]]
    assert k[0].tolist() == [[6.0, 26.0], [6.0, 24.0], [0.0, 24.0], [0.0, 26.0]]
    rng = np.random.default_rng(0)
    cen = rng.uniform(5, 25, (50, 3))
    wlh = rng.uniform(1, 6, (50, 3))
    yaw = rng.uniform(-np.pi, np.pi, 50)
    raw = boxes.bottom_corners_xy(cen, wlh, yaw)
    ci, ki = boxes.boxes_to_image_space(cen, wlh, yaw, 30)

    def signed_area(q):
        x, y = q[:, 0], q[:, 1]
        return 0.5 * (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    for i in range(50):
        assert abs(signed_area(raw[i]) - wlh[i, 0] * wlh[i, 1]) < 1e-9        # CCW before the flip
        assert abs(signed_area(ki[i]) + wlh[i, 0] * wlh[i, 1]) < 1e-9        # CW after it
        assert np.allclose(ki[i].mean(0), ci[i, :2])                          # centre of the quad
    assert np.array_equal(ci[:, 0], cen[:, 0]) and np.array_equal(ci[:, 2], cen[:, 2])
    assert np.allclose(ci[:, 1], 29 - cen[:, 1])


def test_product_anchors_agree_with_the_loop_restatement(oracle):
    """The product's vectorised geometry against the oracle's restatement, which follows the
    reference loop for loop in scalar arithmetic (oracle.py: make_anchor_boxes,
    _one_box_bottom_corners_xy) -- two derivations of the same published formula."""
    from pp_amd import boxes
    cfg = boxes.AnchorConfig.reference_default()
    cfg = boxes.AnchorConfig(7, 5, cfg.fm_scale, cfg.dims, cfg.yaws_deg, cfg.zs)   # all six anchor types
    mine = boxes.make_anchors(cfg)
    corners, centers, wlh, yaw = oracle.make_anchor_boxes(7, 5, 0.5, cfg.dims, cfg.yaws_deg, cfg.zs)
    assert np.allclose(mine["corners"], corners, rtol=0, atol=1e-12)
    assert np.array_equal(mine["centers"], centers) and np.array_equal(mine["wlh"], wlh)
    assert np.allclose(mine["yaw"], yaw, rtol=0, atol=1e-15)
    assert np.allclose(mine["xy"], oracle.anchor_xy_rows(corners, cfg.yaws_deg), rtol=0, atol=1e-12)
    c1, k1 = boxes.boxes_to_image_space([[3., 4, 1], [9, 2, 0.5]], [[2., 5, 1], [1, 3, 2]], [0.7, -2.1], 30)
    c2, k2 = oracle.boxes_to_image_space([[3., 4, 1], [9, 2, 0.5]], [[2., 5, 1], [1, 3, 2]], [0.7, -2.1], 30)
    assert np.array_equal(c1, c2) and np.allclose(k1, k2, rtol=0, atol=1e-12)


def test_synthetic_cloud_statistics(oracle):
    """SURVEY 8d probe: ~56.7k in-range points, ~25.3k cells (> P: overflow regime)."""
    from pp_amd import synth
    from util import grid_args
    pts = synth.lidar_like(60000, 50.0, 0)
    assert pts.dtype == np.float32 and pts.shape == (60000, 4)
    assert np.array_equal(pts, synth.lidar_like(60000, 50.0, 0))          # deterministic
    cc = oracle.cell_counts(pts.astype(np.float64), *grid_args(50.0, 0.2))
    assert 54000 < cc[:, 2].sum() < 59000 and 23000 < len(cc) < 28000 and cc[:, 2].max() < 64


def test_sweep_sharding_partition():
    from pp_amd import shard
    for n, w in ((8, 8), (8, 4), (10, 4), (3, 8), (0, 2)):
        parts = [shard.sweeps_for_rank(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert shard.sweeps_for_rank(8, 3, 8) == [3]                 # one sweep per GPU (config 4)
