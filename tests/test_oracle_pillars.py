"""Pins the CPU oracle of create_pillars against the behavioural probes recorded
in SURVEY.md 5.9 / 8c (the reference ships no tests or fixtures of its own)."""
import numpy as np
import pytest

from util import grid_args


def v1_points():
    return np.array([[1.1, 2.1, 0.0, 10], [0.5, 0.5, 0, 1], [1.2, 2.2, 0.1, 11], [1.3, 2.3, 0.2, 12],
                     [4.0, 1, 0, 0], [1, 1, 1.0, 0], [-0.001, 1, 0, 0], [1.5, 2.5, 0.3, 13],
                     [0.25, 0.75, 0.5, 2], [1.8, 2.8, 0.4, 14]])


V1_ARGS = (3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)   # N=3, P=4, 4x4 canvas, z in [-1,1)


def test_v1_hand_case(oracle):
    """5 points in cell (1,2), 2 in (0,0), 3 rejects, N=3."""
    T, I = np.zeros((4, 3, 9)), np.zeros((4, 3))
    m = oracle.create_pillars(v1_points(), T, I, *V1_ARGS)
    assert m == 2
    # row-major order: (col 1,row 1) before (col 0,row 3); y is flipped into rows
    assert I.tolist() == [[1, 1, 1], [1, 0, 3], [0, 0, 0], [0, 0, 0]]
    # the row printed by the survey probe; means are over all 5 points, not the first 3
    assert np.allclose(T[0, 0], [1.1, 2.1, 0, 10, -0.1, -1.1, 0.28, 0.28, 0.2], atol=1e-12)
    assert np.allclose(T[0, :, 0], [1.1, 1.2, 1.3]) and np.allclose(T[0, :, 3], [10, 11, 12])
    # SURVEY 5.9-7 probe: pt (0.5,0.5) on a 4-row canvas -> yp = 3 - 0.5 = 2.5
    assert T[1, 0, 5] == 2.5 and T[1, 0, 4] == -0.5
    assert not T[1, 2].any() and not T[2:].any()


def test_running_mean_is_sequential_not_two_pass(oracle):
    """pillars.cpp:311-328: m <- m*(n/(n+1)) + v/(n+1) in input order."""
    rng = np.random.default_rng(0)
    pts = np.zeros((50, 4))
    pts[:, :3] = 0.5 + rng.random((50, 3)) * 0.4
    T, I = np.zeros((1, 50, 9)), np.zeros((1, 3))
    oracle.create_pillars(pts, T, I, 50, 1, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    m = pts[0, :3].copy()
    for n in range(1, 50):
        m = m * (n / (n + 1)) + pts[n, :3] / (n + 1)
    assert np.array_equal(T[0, :, 6:9], m[None, :] - pts[:, :3])     # bit-exact


def test_half_open_boundaries(oracle):
    """x_min <= x < x_max on every axis."""
    e = np.nextafter
    pts = np.array([[0.0, 1, 0, 1], [e(4.0, 0), 1, 0, 2], [4.0, 1, 0, 3], [1, 0.0, 0, 4], [1, 4.0, 0, 5],
                    [1, 1, -1.0, 6], [1, 1, 1.0, 7], [1, 1, e(1.0, 0), 8]])
    T, I = np.zeros((8, 4, 9)), np.zeros((8, 3))
    m = oracle.create_pillars(pts, T, I, 4, 8, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    kept = sorted(T[..., 3][T[..., 3] != 0].tolist())
    assert kept == [1, 2, 4, 6, 8] and m == 4


def test_never_zeroes_and_partial_write_on_index_error(oracle):
    T, I = np.full((4, 3, 9), 9.0), np.full((4, 3), 9.0)
    oracle.create_pillars(v1_points(), T, I, *V1_ARGS)
    assert (T[1, 2] == 9.0).all() and (T[2:] == 9.0).all() and (I[2:] == 9.0).all()
    T, I = np.zeros((1, 3, 9)), np.zeros((4, 3))
    with pytest.raises(IndexError):
        oracle.create_pillars(v1_points(), T, I, *V1_ARGS)
    assert T[0].any() and I[0].tolist() == [1, 1, 1] and not I[1:].any()


def test_non_f64_outputs_rejected_and_strided_inputs(oracle):
    with pytest.raises(TypeError):
        oracle.create_pillars(v1_points(), np.zeros((4, 3, 9), np.float32), np.zeros((4, 3)), *V1_ARGS)
    pts = np.asfortranarray(v1_points())              # F-order, like dataset.py:88
    big = np.zeros((4, 3, 18))
    T, I = big[..., ::2], np.zeros((4, 3))
    oracle.create_pillars(pts, T, I, *V1_ARGS)
    Tr, Ir = np.zeros((4, 3, 9)), np.zeros((4, 3))
    oracle.create_pillars(v1_points(), Tr, Ir, *V1_ARGS)
    assert np.array_equal(T, Tr) and np.array_equal(I, Ir) and not big[..., 1::2].any()


def test_orders_agree_as_sets_and_overflow_subset_rule(oracle):
    """Same pillar contents whatever the emission order; under overflow every
    emitted pillar equals the uncapped pillar of that cell (SURVEY 7 hard part 1)."""
    import pp_amd.synth as synth
    from util import pillars_by_cell
    pts = synth.lidar_like(6000, 8.0, 1).astype(np.float64)
    g = grid_args(8.0, 0.2)
    outs = {}
    for order in (oracle.ORDER_ROW_MAJOR, oracle.ORDER_SCRAMBLED, oracle.ORDER_HASH):
        p, i, m = oracle.dataset_voxel_stage(pts, 6000, 16, *g, order=order)
        outs[order] = pillars_by_cell(p, i)
        assert len(outs[order]) == m
    ref = outs[oracle.ORDER_ROW_MAJOR]
    for order, d in outs.items():
        assert d.keys() == ref.keys()
        assert all(np.array_equal(d[k], ref[k]) for k in ref)
    cc = oracle.cell_counts(pts, *g)
    assert len(cc) == len(ref)
    assert np.minimum(cc[:, 2], 16).sum() == sum(int((v[3] != 0).sum()) for v in ref.values())
    for order in (oracle.ORDER_ROW_MAJOR, oracle.ORDER_SCRAMBLED, oracle.ORDER_HASH):
        p, i, m = oracle.dataset_voxel_stage(pts, 500, 16, *g, order=order)
        capped = pillars_by_cell(p, i)
        assert len(capped) == 500 and m == len(ref)
        assert all(np.array_equal(v, ref[k]) for k, v in capped.items())
    # row-major keeps the first P cells in (row, col) order
    p, i, m = oracle.dataset_voxel_stage(pts, 500, 16, *g, order=oracle.ORDER_ROW_MAJOR)
    assert np.array_equal(i[:, 1:], cc[:500, :2])


def test_scramble_mult_is_a_bijection(oracle):
    from math import gcd
    for n in (1, 2, 3, 10, 10201, 251001, 1002001):
        m = oracle.scramble_mult(n)
        assert gcd(m, n) == 1 or n <= 2


def test_grid_dims_bound(oracle):
    nx, ny = oracle.grid_dims(.2, .2, -60, -60, 60, 60)
    assert nx >= 600 and ny >= 600
    assert np.floor((np.nextafter(60.0, 0) - -60) / .2) < nx
    with pytest.raises(ValueError):
        oracle.grid_dims(0, .2, -60, -60, 60, 60)


def test_pybind11_front_of_the_oracle_equals_the_ctypes_one(oracle):
    """oracle/oracle_module.cpp: the reference's module surface (pillars.cpp:429-435: two names, positional signatures,
    None returned, outputs mutated in place) on the oracle's reference-style C loops -- what bench.py times as the CPU
    baseline "through the same pybind11 signatures" (BASELINE.md section 4).  Same arrays as the ctypes front, bit for bit."""
    m = oracle.pybind_module()
    assert sorted(n for n in dir(m) if not n.startswith("_")) == ["create_pillars", "make_ious"]
    rng = np.random.default_rng(5)
    agg = rng.uniform(-9, 9, (4, 20000))
    agg[2] = rng.uniform(-2, 2, 20000)
    pts = agg.transpose([1, 0])                       # the strided view of data/dataset.py:88
    P, N = 3000, 16
    args = (N, P, .25, .25, -8, -8, -3, 8, 8, 3, 64)
    T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
    assert m.create_pillars(pts, T, I, *args) is None
    T2, I2 = np.zeros((P, N, 9)), np.zeros((P, 3))
    oracle.create_pillars(pts, T2, I2, *args, order=oracle.ORDER_HASH)
    assert np.array_equal(T, T2) and np.array_equal(I, I2) and I[:, 0].sum() > 100
    with pytest.raises(IndexError):
        m.create_pillars(pts, np.zeros((10, N, 9)), np.zeros((10, 3)), *args)      # undersized outputs: like .at()
    # make_ious: a unit-square anchor against itself (clockwise ground truth) and a far one
    a = np.array([[[0, 0], [1, 0], [1, 1], [0, 1]]], float)
    g = np.array([[[0, 0], [0, 1], [1, 1], [1, 0]], [[50, 50], [50, 51], [51, 51], [51, 50]]], float)
    out = np.full((1, 2), -7.0)
    assert m.make_ious(a, g, np.array([[.5, .5, 0]]), np.array([[.5, .5, 0], [50.5, 50.5, 0]]), out) is None
    assert out[0, 0] == 1.0 and out[0, 1] == 0.0


def _blocks_by_cell(T, I):
    return {(I[p, 1], I[p, 2]): T[p] for p in np.nonzero(I[:, 0])[0]}


def test_baseline_faithful_variant_equals_the_oracle_bit_for_bit(oracle):
    """oracle/faithful_module.cpp: the CPU baseline BASELINE.md section 4 promises -- pybind11 .at() / .mutable_at() for
    every element, one heap node per in-range point, two std::unordered_map keyed on the cell's doubles (the cost structure
    of pillars.cpp:268-329, 335-396, 416-425) -- is what bench.py reports as cpu_baseline.faithful.  It must emit the
    oracle's bits: per cell the same [N, 9] block and the same index row (the emission ORDER is its hash map's, arbitrary
    like the reference's), the cap and the overflow rule included, and the reference's error behaviour (IndexError from
    the bounds-checked store, pillars.cpp:48-56)."""
    f = oracle.faithful_module()
    assert sorted(n for n in dir(f) if not n.startswith("_")) == ["create_pillars", "make_ious"]
    import pp_amd.synth as synth
    pts = synth.lidar_like(20000, 12.0, 3).astype(np.float64)
    agg = np.ascontiguousarray(pts.T)
    view = agg.transpose([1, 0])                               # the strided view data/dataset.py:88 passes
    for step, y_step, N in ((0.25, 0.25, 8), (0.5, 0.4, 40), (3.0, 3.0, 100)):   # the last: cells far beyond N points
        args = (step, y_step, -12.0, -12.0, -3.0, 12.0, 12.0, 3.0, int(24 / y_step))
        P = 20000
        T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
        assert f.create_pillars(view, T, I, N, P, *args) is None
        T2, I2 = np.zeros((P, N, 9)), np.zeros((P, 3))
        m = oracle.create_pillars(pts, T2, I2, N, P, *args, order=oracle.ORDER_ROW_MAJOR)
        got, want = _blocks_by_cell(T, I), _blocks_by_cell(T2, I2)
        assert len(want) == m and got.keys() == want.keys()
        assert all(np.array_equal(got[k], want[k]) for k in want)
        assert int(I[:, 0].sum()) == m and not T[m:].any()
        # overflow: exactly `cap` pillars, each the uncapped block of its cell; nothing beyond the cap is touched
        cap = m // 3
        Tc, Ic = np.zeros((P, N, 9)), np.zeros((P, 3))
        f.create_pillars(view, Tc, Ic, N, cap, *args)
        capped = _blocks_by_cell(Tc, Ic)
        assert len(capped) == cap and not Tc[cap:].any() and not Ic[cap:].any()
        assert all(np.array_equal(v, want[k]) for k, v in capped.items())
    # strided outputs, never zeroed
    big = np.full((50, 4, 18), 7.0)
    Ts, Is = big[..., ::2], np.full((50, 3), 7.0)
    f.create_pillars(v1_points(), Ts, Is, *V1_ARGS)
    Tr, Ir = np.full((50, 4, 9), 7.0), np.full((50, 3), 7.0)
    oracle.create_pillars(v1_points(), Tr, Ir, *V1_ARGS)
    assert _blocks_by_cell(Ts, Is).keys() == _blocks_by_cell(Tr, Ir).keys() and (big[..., 1::2] == 7.0).all()
    assert all(np.array_equal(v, _blocks_by_cell(Tr, Ir)[k]) for k, v in _blocks_by_cell(Ts, Is).items())
    assert (Ts[2:] == 7.0).all() and (Ts[:2, 3] == 7.0).all()
    # undersized outputs: the checked store raises, as in the reference
    with pytest.raises(IndexError):
        f.create_pillars(view, np.zeros((10, 4, 9)), np.zeros((10, 3)), 8, 2000, 0.25, 0.25, -12.0, -12.0, -3.0, 12.0, 12.0, 3.0, 96)
    with pytest.raises(IndexError):
        f.create_pillars(view, np.zeros((2000, 4, 9)), np.zeros((2000, 3)), 8, 2000, 0.25, 0.25, -12.0, -12.0, -3.0, 12.0, 12.0, 3.0, 96)


def test_baseline_faithful_make_ious_equals_the_oracle(oracle):
    """make_ious of the faithful variant (four checked reads per pair for the gate, a checked store, heap-backed rings for
    a surviving pair; pillars.cpp:416-425, 149-160) against the oracle's: the dense matrix bit for bit, strided and
    transposed outputs, every entry written, wrong winding raised."""
    from pp_amd import boxes, synth
    f = oracle.faithful_module()
    anchors = boxes.make_anchors(boxes.AnchorConfig(60, 60))
    gt = synth.gt_boxes(25, 120, 4, margin=10.0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 120)
    A = anchors["corners"].shape[0]
    want = np.full((A, 25), -3.0)
    oracle.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, want)
    got = np.full((A, 25), -5.0)
    assert f.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, got) is None
    assert np.array_equal(got, want) and (want > 0).sum() > 100
    tr = np.full((25, A), -5.0)
    f.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, tr.T)
    assert np.array_equal(tr.T, want)
    wide = np.full((A, 50), -5.0)
    f.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, wide[:, ::2])
    assert np.array_equal(wide[:, ::2], want) and (wide[:, 1::2] == -5.0).all()
    with pytest.raises(ValueError):
        f.make_ious(anchors["corners"][:, ::-1], k_img, anchors["centers"], c_img, got)
    with pytest.raises(IndexError):
        f.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, np.zeros((A, 24)))
