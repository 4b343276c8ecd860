"""create_target / make_target restatement (utils/box_utils.py:162-232, 70-109):
the exact post-IoU semantics of SURVEY 5.9, on hand-built cases (V6)."""
import numpy as np


def _anchors(O, xs, w=4.0, l=8.0):
    centers = np.array([[x, 20.0, 0.5] for x in xs])
    wlh = np.tile([w, l, 1.5], (len(xs), 1))
    yaw = np.zeros(len(xs))
    return O.box_bottom_corners_xy(centers, wlh, yaw), centers, wlh, yaw


def _gts(O, rows, H):
    c = np.array([[r[0], r[1], 0.7] for r in rows], float)
    wlh = np.array([[r[2], r[3], 1.6] for r in rows], float)
    yaw = np.array([r[4] for r in rows], float)
    cls = np.array([r[5] for r in rows])
    c_img, k_img = O.boxes_to_image_space(c, wlh, yaw, H)
    return c, wlh, yaw, cls, c_img, k_img


def _run(O, anchors, gts, H, thresh=0.6):
    ac, an, aw, ay = anchors
    c, wlh, yaw, cls, c_img, k_img = gts
    return O.create_target(ac, k_img, an, c_img, aw, ay, c, wlh, yaw, cls, H, pos_thresh=thresh)


def test_positive_strict_threshold_and_forced_anchor(oracle):
    O, H = oracle, 41                      # (H-1) - 20 = 20: gt rows land on the anchors' row
    anchors = _anchors(O, [12.0, 20.0, 23.0, 40.0])
    # gt0 == anchor 1 (IoU 1); gt1 far from everything except a weak overlap with anchor 3
    gts = _gts(O, [(20.0, 20.0, 4, 8, 0.0, 3), (45.0, 20.0, 4, 8, 0.0, 5)], H)
    cls_t, reg_t, ious = _run(O, anchors, gts, H)
    assert ious[1, 0] == 1.0 and 0 < ious[3, 1] < 0.6
    assert cls_t[1].tolist() == [0, 0, 0, 1, 0, 0, 0, 0, 0]          # positive by threshold
    assert cls_t[3].tolist() == [0, 0, 0, 0, 0, 1, 0, 0, 0]          # forced: best anchor of gt1
    assert reg_t[1, 0] == 1 and reg_t[3, 0] == 1 and not cls_t[[0, 2]].any() and not reg_t[[0, 2]].any()
    assert abs(reg_t[1, 1]) < 1e-12 and abs(reg_t[1, 4] - np.log(4 / 4)) < 1e-12   # dx, dw of the exact match
    # IoU exactly at the threshold is NOT positive (strict >, box_utils.py:195)
    exact = ious[2, 0]
    cls2, reg2, _ = _run(O, anchors, gts, H, thresh=exact)
    assert not cls2[2].any()
    cls3, _, _ = _run(O, anchors, gts, H, thresh=np.nextafter(exact, 0))
    assert cls3[2, 3] == 1


def test_gt_whose_best_anchor_is_index0_is_dropped(oracle):
    """np.nonzero filter, box_utils.py:204-205: argmax == 0 means "no anchor"."""
    O, H = oracle, 41
    anchors = _anchors(O, [20.0, 40.0])
    gts = _gts(O, [(23.5, 20.0, 4, 8, 0.0, 2)], H)        # overlaps anchor 0 only, IoU < 0.6
    cls_t, reg_t, ious = _run(O, anchors, gts, H)
    assert 0 < ious[0, 0] < 0.6
    assert not cls_t.any() and not reg_t.any()


def test_duplicate_forced_anchor_sets_both_classes_last_reg_wins(oracle):
    O, H = oracle, 41
    anchors = _anchors(O, [5.0, 20.0])
    gts = _gts(O, [(22.5, 20.0, 4, 8, 0.0, 1), (17.5, 20.0, 4, 8, 0.0, 6)], H)
    cls_t, reg_t, ious = _run(O, anchors, gts, H)
    assert (ious[1] > 0).all() and (ious[1] < 0.6).all()
    assert cls_t[1].tolist() == [0, 1, 0, 0, 0, 0, 1, 0, 0]
    exp = O.make_target(anchors[1][1], anchors[2][1], anchors[3][1], gts[0][1], gts[1][1], gts[2][1], H)
    assert reg_t[1].tolist() == exp            # the last ground truth wins (box_utils.py:223-228)


def test_make_target_formulae(oracle):
    """box_utils.py:70-109: yaw folding into (-pi/2, pi/2), ort by quadrant, y flip."""
    O = oracle
    H = 100
    a_c, a_wlh = (10.0, 20.0, 0.5), (3.0, 4.0, 2.0)
    r = O.make_target(a_c, a_wlh, 0.0, (13.0, 70.0, 1.5), (6.0, 2.0, 1.0), 3.0, H)
    ad = 5.0
    assert r[0] == 1 and abs(r[1] - 3 / ad) < 1e-15 and abs(r[2] - ((99 - 70) - 20) / ad) < 1e-15
    assert abs(r[3] - 0.5) < 1e-15 and abs(r[4] - np.log(2)) < 1e-15 and abs(r[5] - np.log(.5)) < 1e-15
    assert abs(r[7] - np.sin(3.0 - np.pi)) < 1e-15 and r[8] == 0          # folded: |gt-at| < pi/2
    r = O.make_target(a_c, a_wlh, np.pi / 2, (13.0, 70.0, 1.5), (6.0, 2.0, 1.0), -0.2, H)
    assert abs(r[7] - np.sin(-0.2 - np.pi / 2)) < 1e-15 and r[8] == 1     # gt-at in [-pi,-pi/2]
    r = O.make_target(a_c, a_wlh, 0.0, (13.0, 70.0, 1.5), (6.0, 2.0, 1.0), -2.0, H)
    assert abs(r[7] - np.sin(-2.0 + np.pi)) < 1e-15 and r[8] == 0


def test_anchor_grid_layout(oracle):
    """box_utils.py:133-150: anchors ordered (y, x, d); CCW corners; gt CW after flip."""
    O = oracle
    corners, centers, wlh, yaw = O.make_anchor_boxes(3, 4, 0.5, [[10, 25, 1.75]] * 2, [0, 90], [.75, .75])
    assert corners.shape == (24, 4, 2)
    i = (2 * 4 + 1) * 2 + 1            # y=2, x=1, d=1
    assert centers[i].tolist() == [3.0, 5.0, 0.75] and abs(yaw[i] - np.pi / 2) < 1e-15

    def area(q):
        return 0.5 * sum(q[k, 0] * q[(k + 1) % 4, 1] - q[(k + 1) % 4, 0] * q[k, 1] for k in range(4))
    assert all(area(c) > 0 for c in corners)                       # counter-clockwise
    assert abs(area(corners[0]) - 250) < 1e-9
    c_img, k_img = O.boxes_to_image_space([[5., 7, 1]], [[2., 4, 1]], [0.3], 20)
    assert area(k_img[0]) < 0 and c_img[0, 1] == 12.0              # clockwise after the y flip


def test_lidar_ingest_restatement(oracle):
    """dataset.py:65-88 via the (recalled) LidarPointCloud semantics: first four f32
    columns, f64 rigid transform stored as f32, remove_close needs BOTH |x| and |y|
    inside the radius, sweeps are concatenated in order."""
    raw = np.array([[1, 2, 3, 9, 7], [0.0005, -0.0005, 5, 8, 7], [0.0005, 2, 0, 6, 7], [0.5, 0.25, -1, 4, 7]],
                   np.float32)
    out = oracle.lidar_ingest([(raw, np.eye(4))])
    assert out.shape == (3, 4) and out[:, 3].tolist() == [9, 6, 4]        # row 1 removed only
    shift = np.eye(4)
    shift[:3, 3] = [10, 20, 30]
    rot = np.eye(4)
    rot[:2, :2] = [[0, -1], [1, 0]]
    out2 = oracle.lidar_ingest([(raw, shift), (raw[:1], rot)])
    assert out2.shape == (5, 4)
    assert out2[0].tolist() == [11, 22, 33, 9] and out2[4].tolist() == [-2, 1, 3, 9]
    assert out2.dtype == np.float64 and np.array_equal(out2, out2.astype(np.float32))


def test_nms_and_postprocess_restatement(oracle):
    """torchvision.ops.nms semantics (greedy, IoU > thresh suppresses, score order)
    and the decode formulae of make_pred_boxes / move_box_to_car_space."""
    O = oracle
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], np.float32)
    keep = O.nms(b, np.array([0.9, 0.8, 0.7, 0.95], np.float32), 0.5)
    assert keep.tolist() == [3, 2]              # box 3 wins, suppresses 0 (IoU .95) and 1 (.66); 2 is far
    assert O.nms(b, np.array([0.9, 0.8, 0.7, 0.95], np.float32), 0.99).tolist() == [3, 0, 1, 2]
    # one anchor, hand-computed decode
    a_c, a_wlh, a_yaw = np.array([[30.0, 40.0, 0.75]]), np.array([[10.0, 25.0, 1.75]]), np.array([0.0])
    corners = O.box_bottom_corners_xy(a_c, a_wlh, a_yaw)
    xy = O.anchor_xy(corners, [0])
    assert xy[0].tolist() == [17.5, 45.0, 42.5, 35.0]            # corners 2 and 0 (box_utils.py:155)
    cls = np.full((9, 1, 1), -5.0, np.float32)
    cls[4] = 2.0
    reg = np.zeros((8, 1, 1), np.float32)
    reg[[0, 3, 6], 0, 0] = [0.1, np.log(2.0), 0.5]
    out, kept = O.postprocess(cls, reg, a_c, a_wlh, a_yaw, xy, 100, 0.2, 0.2, -10, -10)
    diag = np.sqrt(10.0 ** 2 + 25.0 ** 2)
    assert kept.tolist() == [0] and out[0, 8] == 4 and abs(out[0, 7] - 1 / (1 + np.exp(-2.0))) < 1e-6
    assert abs(out[0, 0] - ((30 + np.float32(0.1) * diag) * 0.2 - 10)) < 1e-9
    assert abs(out[0, 1] - ((99 - 40.0) * 0.2 - 10)) < 1e-9
    assert abs(out[0, 3] - 2.0 * 10 * 0.2) < 1e-5 and abs(out[0, 6] - np.arcsin(np.tanh(np.float32(0.5)))) < 1e-6
