"""Pins the CPU oracle of make_ious / iou (data/pillars.cpp:132-172, 400-427).
Boost.Geometry is absent: the known answers are analytic."""
import numpy as np
import pytest


def rect(O, cx, cy, w, l, yaw, cw=False):
    q = O.box_bottom_corners_xy(np.array([cx, cy, 0.]), np.array([w, l, 1.]), np.array(yaw))
    return q[[0, 3, 2, 1]].copy() if cw else q


def test_known_answers(oracle):
    O = oracle
    a = rect(O, 5, 5, 2, 4, 0.0)
    assert O.iou_pair(a, rect(O, 5, 5, 2, 4, 0.0, cw=True)) == 1.0
    assert abs(O.iou_pair(a, rect(O, 6, 5, 2, 4, 0.0, cw=True)) - 0.6) < 1e-12     # shifted 1 along length
    assert abs(O.iou_pair(a, rect(O, 5, 5, 2, 4, np.pi / 2, cw=True)) - 1 / 3) < 1e-12
    a4 = rect(O, 5, 5, 4, 4, 0.0)
    inter = 16 * (2 * np.sqrt(2) - 2)
    assert abs(O.iou_pair(a4, rect(O, 5, 5, 4, 4, np.pi / 4, cw=True)) - inter / (32 - inter)) < 1e-12
    assert O.iou_pair(a, rect(O, 50, 5, 2, 4, 0.3, cw=True)) == 0.0              # disjoint
    assert O.iou_pair(a, rect(O, 9, 5, 2, 4, 0.0, cw=True)) == 0.0                # touching edge only


def test_wrong_winding_raises_instead_of_exit(oracle):
    a = rect(oracle, 5, 5, 2, 4, 0.2)
    with pytest.raises(ValueError):
        oracle.iou_pair(a, a)                      # gt given counter-clockwise
    with pytest.raises(ValueError):
        oracle.iou_pair(a[::-1].copy(), rect(oracle, 5, 5, 2, 4, 0.2, cw=True))


def test_gate_and_layout(oracle):
    """pillars.cpp:418-424: strict > 10 gate on |dx|, |dy|; every entry written."""
    O = oracle
    ac = np.stack([rect(O, x, 20, 4, 8, 0.0) for x in (20.0, 30.0, 31.0, 20.0)])
    an = np.array([[20., 20, 0], [30, 20, 0], [31, 20, 0], [20, 31, 0]])
    gc = np.stack([rect(O, 20, 20, 4, 8, 0.0, cw=True), rect(O, 24, 20, 4, 8, 0.0, cw=True)])
    gn = np.array([[20., 20, 0], [24, 20, 0]])
    ious = np.full((4, 2), -1.0)
    O.make_ious(ac, gc, an, gn, ious)
    assert ious[0, 0] == 1.0 and abs(ious[0, 1] - 1 / 3) < 1e-12   # overlap 4 of 8+8-4
    assert ious[1, 0] == 0.0                 # centres exactly 10 apart: not gated, disjoint
    assert ious[2, 0] == 0.0 and ious[3, 0] == 0.0   # 11 apart in x / in y: gated
    assert (ious >= 0).all()
    # gate uses the centres, not the corners: lie about a centre and the pair is skipped
    an2 = an.copy()
    an2[0, 0] = 40.0
    O.make_ious(ac, gc, an2, gn, ious)
    assert ious[0, 0] == 0.0


def test_random_pairs_against_monte_carlo_and_symmetry(oracle):
    O = oracle
    rng = np.random.default_rng(5)
    for _ in range(40):
        pa = (rng.uniform(8, 12), rng.uniform(8, 12), rng.uniform(2, 5), rng.uniform(3, 9), rng.uniform(-np.pi, np.pi))
        pg = (rng.uniform(8, 12), rng.uniform(8, 12), rng.uniform(2, 5), rng.uniform(3, 9), rng.uniform(-np.pi, np.pi))
        v = O.iou_pair(rect(O, *pa), rect(O, *pg, cw=True))
        # roles swapped: identical up to rounding
        assert abs(v - O.iou_pair(rect(O, *pg), rect(O, *pa, cw=True))) < 1e-12
        # rigid motion invariance
        th, tx, ty = rng.uniform(-3, 3), rng.uniform(-5, 5), rng.uniform(-5, 5)

        def move(p):
            c, s = np.cos(th), np.sin(th)
            return (c * p[0] - s * p[1] + tx, s * p[0] + c * p[1] + ty, p[2], p[3], p[4] + th)
        assert abs(v - O.iou_pair(rect(O, *move(pa)), rect(O, *move(pg), cw=True))) < 1e-9
        # independent estimate: area by point sampling
        xs = rng.uniform(0, 20, (200000, 2))

        def inside(p):
            c, s = np.cos(p[4]), np.sin(p[4])
            dx, dy = xs[:, 0] - p[0], xs[:, 1] - p[1]
            u, w = c * dx + s * dy, -s * dx + c * dy
            return (np.abs(u) <= p[3] / 2) & (np.abs(w) <= p[2] / 2)
        ia, ig = inside(pa), inside(pg)
        mc = (ia & ig).sum() / max(1, (ia | ig).sum())
        assert abs(v - mc) < 0.02
