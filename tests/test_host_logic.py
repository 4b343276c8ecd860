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


def test_product_anchors_equal_oracle_anchors(oracle):
    from pp_amd import boxes
    cfg = boxes.AnchorConfig(7, 5)
    mine = boxes.make_anchors(cfg)
    corners, centers, wlh, yaw = oracle.make_anchor_boxes(7, 5, 0.5, cfg.dims, cfg.yaws_deg, cfg.zs)
    assert np.array_equal(mine["corners"], corners) and np.array_equal(mine["centers"], centers)
    assert np.array_equal(mine["wlh"], wlh) and np.array_equal(mine["yaw"], yaw)
    c1, k1 = boxes.boxes_to_image_space([[3., 4, 1]], [[2., 5, 1]], [0.7], 30)
    c2, k2 = oracle.boxes_to_image_space([[3., 4, 1]], [[2., 5, 1]], [0.7], 30)
    assert np.array_equal(c1, c2) and np.array_equal(k1, k2)


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
