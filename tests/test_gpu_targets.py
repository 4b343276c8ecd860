"""GPU parity tests of the rotated-IoU and fused anchor-target kernels against
the CPU oracle (create_target, utils/box_utils.py:162-232).

Bar: IoU bit-exact (same f64 operation sequence, contraction off); class targets
exact (integer work); regression targets within 1e-6 of the f32-cast oracle
(log/sin come from different libms)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REG_TOL = 1e-6


def _oracle_targets(O, anchors, gt, H, thresh=0.6):
    # the SAME image-space box arrays the device path derives (pp_amd.boxes, pinned by hand-computed
    # values and against the oracle's loop restatement in tests/test_host_logic.py): the bit-exact
    # comparisons below are about the kernels, not about 1-ulp differences of two cos() routines
    from pp_amd import boxes
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
    return O.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                           anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                           pos_thresh=thresh)


def _check(cls_t, reg_t, ref_c, ref_r):
    cls_t, reg_t = cls_t.cpu().numpy(), reg_t.cpu().numpy()
    assert np.array_equal(cls_t, ref_c.astype(np.float32)), "class targets differ"
    assert np.array_equal(reg_t[:, 0], ref_r[:, 0].astype(np.float32)), "positive flags differ"
    assert np.array_equal(reg_t[:, 8], ref_r[:, 8].astype(np.float32)), "orientation bits differ"
    assert np.abs(reg_t - ref_r.astype(np.float32)).max() <= REG_TOL


def test_config3_full_size(gpu, oracle):
    """BASELINE config 3: 250x250 feature map x 2 anchors = 125 000 anchors, G = 40."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
    assert len(anchors["corners"]) == 125000
    ta = TargetAssigner(anchors, canvas_height=500, device=gpu)
    for seed in (0, 1):
        gt = synth.gt_boxes(40, 500, seed)
        cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
        torch.cuda.synchronize()
        ref_c, ref_r, ious = _oracle_targets(oracle, anchors, gt, 500)
        assert (ref_r[:, 0] == 1).sum() > 40            # there are positives beyond the forced ones
        _check(cls_t, reg_t, ref_c, ref_r)
        # dense IoU matrix on the device: bit-exact
        c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)
        d = ta.ious(k_img, c_img).cpu().numpy()
        assert np.array_equal(d, ious)


def test_reference_default_anchor_set_sampled(gpu, oracle):
    """config.py default: 6 anchors/cell (3 sizes x 2 yaws); 120x120 map keeps the oracle quick."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    ref = boxes.AnchorConfig.reference_default()
    cfg = boxes.AnchorConfig(120, 120, 0.5, ref.dims, ref.yaws_deg, ref.zs)
    anchors = boxes.make_anchors(cfg)
    ta = TargetAssigner(anchors, canvas_height=240, device=gpu)
    gt = synth.gt_boxes(25, 240, 7, margin=20.0)
    gt["wlh"][::3, :2] = [4.0, 8.0]                    # some small boxes too
    cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    torch.cuda.synchronize()
    ref_c, ref_r, _ = _oracle_targets(oracle, anchors, gt, 240)
    _check(cls_t, reg_t, ref_c, ref_r)


@pytest.mark.parametrize("source", ["arrays", "grid"])
def test_reference_default_anchor_set_full_size(gpu, oracle, source):
    """config.py:55-59,109-116 exactly: 300x300 feature map x 6 anchors = 540 000 anchors on the 600x600
    canvas, G = 40 boxes of all three size groups; uploaded anchor arrays and the on-the-fly grid, against
    the oracle's full [A,G] evaluation."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    cfg = boxes.AnchorConfig.reference_default()
    anchors = boxes.make_anchors(cfg)
    assert len(anchors["corners"]) == 540000
    ta = TargetAssigner(anchors if source == "arrays" else cfg, canvas_height=600, device=gpu)
    gt = synth.gt_boxes(40, 600, 3, margin=60.0)
    gt["wlh"][::3, :2] = boxes.SMALL[:2] * 1.05          # small and large boxes too
    gt["wlh"][1::3, :2] = boxes.LARGE[:2] * 0.95
    cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    torch.cuda.synchronize()
    ref_c, ref_r, ious = _oracle_targets(oracle, anchors, gt, 600)
    assert (ref_r[:, 0] == 1).sum() >= 40 and (ious > 0).sum() > 1000
    _check(cls_t, reg_t, ref_c, ref_r)


def test_quirks_anchor0_duplicates_ties_threshold(gpu, oracle):
    """a ground truth whose best anchor is index 0 is dropped; two
    ground truths sharing one best anchor set both classes, last regression row
    wins; exact IoU ties resolve to the first index; IoU == threshold is not positive."""
    import torch
    from pp_amd.targets import TargetAssigner
    O = oracle
    xs = [20.0, 60.0, 63.0, 100.0, 140.0, 148.0]
    centers = np.array([[x, 20.0, 0.5] for x in xs])
    wlh = np.tile([4.0, 8.0, 1.5], (len(xs), 1))
    yaw = np.zeros(len(xs))
    anchors = {"corners": O.box_bottom_corners_xy(centers, wlh, yaw), "centers": centers,
               "wlh": wlh, "yaw": yaw}
    H = 41
    gt = {"centers": np.array([[23.5, 20, .7],      # best anchor is index 0 -> dropped
                               [102.5, 20, .7], [97.5, 20, .7],   # both forced onto anchor 3
                               [144.0, 20, .7],     # exact tie between anchors 4 and 5 -> first
                               [60.0, 20, .7]]),    # exact match of anchor 1, overlaps anchor 2
          "wlh": np.tile([4.0, 8.0, 1.6], (5, 1)), "yaw": np.zeros(5),
          "classes": np.array([2, 1, 6, 4, 8], np.int32)}
    ta = TargetAssigner(anchors, canvas_height=H, device=gpu)
    cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    torch.cuda.synchronize()
    ref_c, ref_r, ious = _oracle_targets(O, anchors, gt, H)
    assert ious[4, 3] == ious[5, 3] > 0 and not ref_c[0].any()
    assert ref_c[3].tolist() == [0, 1, 0, 0, 0, 0, 1, 0, 0] and ref_c[4, 4] == 1 and not ref_c[5].any()
    _check(cls_t, reg_t, ref_c, ref_r)
    thr = float(ious[2, 4])                            # IoU exactly at the threshold
    ta2 = TargetAssigner(anchors, canvas_height=H, pos_thresh=thr, device=gpu)
    cls2, reg2 = ta2.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
    ref_c2, ref_r2, _ = _oracle_targets(O, anchors, gt, H, thresh=thr)
    assert not ref_c2[2].any()
    _check(cls2, reg2, ref_c2, ref_r2)


def test_no_ground_truth_and_wrong_winding(gpu, oracle):
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    anchors = boxes.make_anchors(boxes.AnchorConfig(20, 20))
    ta = TargetAssigner(anchors, canvas_height=40, device=gpu)
    cls_t, reg_t = ta.assign(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0, np.int32), check=True)
    assert not cls_t.any() and not reg_t.any()
    # ground truth handed over counter-clockwise (not flipped): raises, never exits
    c = np.array([[20.0, 20.0, 0.5]])
    k = boxes.bottom_corners_xy(c, np.array([[10.0, 25.0, 1.7]]), np.array([0.2]))
    with pytest.raises(ValueError):
        ta.ious(k, c)
    torch.cuda.synchronize()


def test_anchor_grid_on_the_fly_equals_uploaded_arrays(gpu, oracle):
    """anchors evaluated in the kernels from (feature-map size, dims, yaws,
    zs) -- no per-anchor arrays -- give the SAME bits as the uploaded make_anchors arrays
    (and therefore the oracle's targets): BASELINE config 3 and the reference's 300x300x6 set."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    for cfg, H, n_gt in ((boxes.AnchorConfig(250, 250), 500, 40),
                         (boxes.AnchorConfig.reference_default(), 600, 25)):
        anchors = boxes.make_anchors(cfg)
        ta_arr = TargetAssigner(anchors, canvas_height=H, device=gpu)
        ta_grid = TargetAssigner(cfg, canvas_height=H, device=gpu)
        assert ta_grid.A == ta_arr.A == cfg.num_anchors and ta_grid.a_corners is None
        for seed in (3, 4):
            gt = synth.gt_boxes(n_gt, H, seed)
            c0, r0 = ta_arr.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
            c1, r1 = ta_grid.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
            torch.cuda.synchronize()
            assert torch.equal(c0, c1) and torch.equal(r0, r1)
            assert (r1[:, 0] == 1).sum().item() > 0
        # no ground truth: all-zero targets from the grid path too
        c1, r1 = ta_grid.assign(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0, np.int32))
        assert not c1.any() and not r1.any()
    # and against the oracle directly (config 3)
    cfg = boxes.AnchorConfig(250, 250)
    anchors = boxes.make_anchors(cfg)
    gt = synth.gt_boxes(40, 500, 9)
    c1, r1 = TargetAssigner(cfg, canvas_height=500, device=gpu).assign(
        gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    ref_c, ref_r, _ = _oracle_targets(oracle, anchors, gt, 500)
    _check(c1, r1, ref_c, ref_r)
    # the type table reproduces make_anchors' corners exactly
    tab = boxes.anchor_type_table(cfg)
    d = np.arange(cfg.num_anchors) % cfg.per_cell
    corners = tab[d, :8].reshape(-1, 4, 2) + anchors["centers"][:, None, :2]
    assert np.array_equal(corners, anchors["corners"])


def _crowded_gt(rng, G, H, n_cluster, spot, classes=9):
    """G boxes, the first n_cluster of them packed around one spot (every anchor there passes the
    centre gate of all of them), the rest spread over the canvas; a few exact duplicates."""
    c = np.column_stack([rng.uniform(5, H - 5, G), rng.uniform(5, H - 5, G), rng.uniform(0.2, 1.5, G)])
    c[:n_cluster, :2] = spot + rng.uniform(-4, 4, (n_cluster, 2))
    gt = {"centers": c,
          "wlh": np.column_stack([rng.uniform(3, 12, G), rng.uniform(6, 26, G), rng.uniform(1, 3, G)]),
          "yaw": rng.uniform(-np.pi, np.pi, G), "classes": rng.integers(0, classes, G).astype(np.int32)}
    for k in ("centers", "wlh", "yaw"):
        gt[k][3] = gt[k][2]
        gt[k][G - 1] = gt[k][5]
    return gt


@pytest.mark.parametrize("fm,G,n_cluster,classes", [(60, 300, 200, 9),      # 5 chunks of 64, ~25 windows of 512 pairs
                                                    (40, 2100, 64, 9),      # tail beyond its LDS (> 2048)
                                                    (50, 70, 30, 20),       # rows wider than the LDS stage
                                                    (50, 40, 20, 70)])      # > 64 classes: general forced path
def test_crowded_scenes_many_boxes_wide_rows(gpu, oracle, fm, G, n_cluster, classes):
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(fm * 1000 + G)
    cfg = boxes.AnchorConfig(fm, fm)
    anchors = boxes.make_anchors(cfg)
    H = 2 * fm
    gt = _crowded_gt(rng, G, H, n_cluster, np.array([0.45 * H, 0.6 * H]), classes)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
    ref_c, ref_r, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                           anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                                           pos_thresh=0.45, num_classes=classes)
    assert (ref_r[:, 0] == 1).sum() > 10
    for src in (cfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=0.45, num_classes=classes, device=gpu)
        for _ in range(2):      # twice: the kernel's tail must have re-armed its scratch words
            cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
        torch.cuda.synchronize()
        _check(cls_t, reg_t, ref_c, ref_r)


def test_targets_into_unaligned_outputs(gpu, oracle):
    """The C ABI takes any float pointer: rows that do not start on 16 bytes take the scalar stores."""
    import ctypes
    import torch
    from pp_amd import _lib, boxes, synth
    from pp_amd.targets import TargetAssigner, _vp
    cfg = boxes.AnchorConfig(64, 64)
    ta = TargetAssigner(cfg, canvas_height=128, device=gpu)
    gt = synth.gt_boxes(12, 128, 5)
    c0, r0 = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    g = ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
    cbuf = torch.full((ta.A * 9 + 3,), 7.0, dtype=torch.float32, device=gpu)
    rbuf = torch.full((ta.A * 9 + 3,), 7.0, dtype=torch.float32, device=gpu)
    stream = ctypes.c_void_p(torch.cuda.current_stream(ta.device).cuda_stream)
    rc = _lib.lib().pp_assign_targets_grid_dev(
        ta._ctx.handle, stream, cfg.fm_height, cfg.fm_width, float(cfg.fm_scale), cfg.per_cell, _vp(ta.types),
        12, *[_vp(x) for x in g], ctypes.byref(ta._prm), _vp(cbuf[1:]), _vp(rbuf[3:]))
    _lib.check(rc, "pp_assign_targets_grid_dev")
    torch.cuda.synchronize()
    assert torch.equal(cbuf[1:1 + ta.A * 9].view(ta.A, 9), c0) and torch.equal(rbuf[3:].view(ta.A, 9), r0)
    assert cbuf[0] == 7 and (cbuf[1 + ta.A * 9:] == 7).all() and (rbuf[:3] == 7).all()
    assert (r0[:, 0] == 1).sum().item() > 0


@pytest.mark.parametrize("fm,per_cell,G,classes", [(24, 10, 7, 9),     # type table beyond its LDS copy (> 8 types)
                                                   (9, 1, 1, 1),       # fewer anchors than one workgroup, one box, one class
                                                   (33, 5, 64, 3),     # exactly one full chunk of ground truths
                                                   (33, 2, 65, 12)])   # one box into the second chunk
def test_small_and_odd_shapes(gpu, oracle, fm, per_cell, G, classes):
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(fm * 100 + per_cell)
    dims = tuple((float(rng.uniform(3, 9)), float(rng.uniform(6, 16)), float(rng.uniform(1, 3))) for _ in range(per_cell))
    yaws = tuple(float(rng.choice([0.0, 90.0, 45.0, 20.0])) for _ in range(per_cell))
    zs = tuple(float(rng.uniform(0.2, 1.5)) for _ in range(per_cell))
    cfg = boxes.AnchorConfig(fm, fm, 0.5, dims, yaws, zs)
    anchors = boxes.make_anchors(cfg)
    H = 2 * fm
    gt = {"centers": np.column_stack([rng.uniform(2, H - 2, G), rng.uniform(2, H - 2, G), rng.uniform(0, 2, G)]),
          "wlh": np.column_stack([rng.uniform(3, 9, G), rng.uniform(6, 16, G), rng.uniform(1, 3, G)]),
          "yaw": rng.uniform(-np.pi, np.pi, G), "classes": rng.integers(0, classes, G).astype(np.int32)}
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
    ref_c, ref_r, ious = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                              anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                                              pos_thresh=0.5, num_classes=classes)
    for src in (cfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=0.5, num_classes=classes, device=gpu)
        cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
        torch.cuda.synchronize()
        _check(cls_t, reg_t, ref_c, ref_r)
    d = TargetAssigner(anchors, canvas_height=H, device=gpu).ious(k_img, c_img).cpu().numpy()
    assert np.array_equal(d, ious)


def _stream(gpu):
    import ctypes
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)


def _soak_single(ta, sets, refs, gpu, calls):
    """Alternating inputs on ONE context: call k runs set order[k]; the first call of every set is checked against
    its OWN oracle result, every later call of the set must reproduce that call's bits (mismatches are counted on the
    device: nothing synchronises inside the loop, so the launches run back to back with the side traffic)."""
    import torch
    from util import SideTraffic, check_targets, soak_order
    g = [ta._gt_to_device(s["centers"], s["wlh"], s["yaw"], s["classes"]) for s in sets]
    order = soak_order(len(sets), calls)
    first, bad = {}, torch.zeros((), dtype=torch.int64, device=gpu)
    traffic = SideTraffic(gpu)
    for it, k in enumerate(order):
        if it % 3 == 0:
            traffic.flush()
        if it % 2 == 0:
            traffic.poke()
        c, r = ta.assign_device(*g[k])
        if k not in first:
            first[k] = (c, r)
        else:
            bad += (torch.ne(c, first[k][0]).any() | torch.ne(r, first[k][1]).any()).to(torch.int64)
    traffic.close()
    torch.cuda.synchronize()
    assert ta._L.pp_iou_check(ta._ctx.handle, _stream(gpu)) == 0
    for k, (c, r) in first.items():
        check_targets(c, r, *refs[k])
    assert int(bad.item()) == 0, f"{int(bad.item())} of {len(order)} calls differ from their own set's result"
    return len(order)


def test_repeated_calls_are_identical(gpu, oracle):
    """The kernels' tail reads what the other workgroups (other XCDs) published through write-through stores and a
    ticket.  A soak that repeats ONE input cannot see a stale read (the previous call left the same bytes), so the
    calls ALTERNATE between six box sets on one context -- different numbers of boxes, positives, forced anchors,
    one empty set, every ordered pair of sets adjacent at least once -- with cache-flushing traffic on the stream and
    unrelated traffic on a second stream; every call is held against its own set's oracle result.  Both kernels
    (k_targets_gt: anchors on the fly; k_targets: uploaded arrays), BASELINE config 3 and the reference default."""
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from util import oracle_targets_for, soak_gt_sets
    for cfg, H in ((boxes.AnchorConfig(250, 250), 500), (boxes.AnchorConfig.reference_default(), 600)):
        anchors = boxes.make_anchors(cfg)
        sets = soak_gt_sets(H)
        refs = [oracle_targets_for(oracle, anchors, s, H, 0.6) for s in sets]
        assert sum(int((r[:, 0] == 1).sum()) for _, r in refs) >= 60
        assert len({int((r[:, 0] == 1).sum()) for _, r in refs}) >= 4       # the sets really differ
        for src in (cfg, anchors):
            ta = TargetAssigner(src, canvas_height=H, device=gpu)
            assert _soak_single(ta, sets, refs, gpu, 300) >= 300


# --------------------------------------------------------------------------- a batch of samples per launch
def _ragged_batch(H, seeds, counts):
    from pp_amd import synth
    out = []
    for s, n in zip(seeds, counts):
        g = synth.gt_boxes(max(n, 1), H, s)
        out.append({k: v[:n] for k, v in g.items()})
    return out


@pytest.mark.parametrize("source", ["arrays", "grid"])
def test_batch_equals_per_sample_calls_and_oracle_config3(gpu, oracle, source):
    """One launch for the B samples of a step (config.py:135: BATCH_SIZE = 4; data/dataset.py:113-118 per sample):
    a ragged batch -- 40, 0, 17 and 33 boxes -- at BASELINE config 3 (A = 125 000) must give, sample by sample, the
    bits of the single-sample call and the oracle's targets.  The list, counter, ticket and column scratch of the
    last-workgroup tail are per sample: a shared ticket would show up here as a missing forced row."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    cfg = boxes.AnchorConfig(250, 250)
    anchors = boxes.make_anchors(cfg)
    ta = TargetAssigner(anchors if source == "arrays" else cfg, canvas_height=500, device=gpu)
    gts = _ragged_batch(500, (0, 1, 2, 3), (40, 0, 17, 33))
    cls_b, reg_b = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    assert cls_b.shape == (4, 125000, 9) and reg_b.shape == (4, 125000, 9)
    for b, g in enumerate(gts):
        c1, r1 = ta.assign(g["centers"], g["wlh"], g["yaw"], g["classes"], check=True)
        assert torch.equal(cls_b[b], c1) and torch.equal(reg_b[b], r1), f"sample {b} differs from its own call"
        if len(g["yaw"]):
            ref_c, ref_r, _ = _oracle_targets(oracle, anchors, g, 500)
            _check(cls_b[b], reg_b[b], ref_c, ref_r)
        else:
            assert not cls_b[b].any() and not reg_b[b].any()
    # the same batch again, and a differently ragged one on the same context (the tails re-armed their words)
    c2, r2 = ta.assign_batch(gts)
    assert torch.equal(c2, cls_b) and torch.equal(r2, reg_b)
    gts2 = [gts[3], gts[0], gts[2]]
    c3, r3 = ta.assign_batch(gts2)
    assert torch.equal(c3[0], cls_b[3]) and torch.equal(c3[1], cls_b[0]) and torch.equal(r3[2], reg_b[2])


def test_batch_reference_default_anchor_set_full_size(gpu, oracle):
    """config.py's shipped anchor set (540 000 anchors) as a batch of four, against the oracle per sample."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    cfg = boxes.AnchorConfig.reference_default()
    anchors = boxes.make_anchors(cfg)
    ta = TargetAssigner(cfg, canvas_height=600, device=gpu)
    gts = _ragged_batch(600, (3, 4, 5, 6), (40, 25, 1, 31))
    gts[0]["wlh"][::3, :2] = boxes.SMALL[:2] * 1.05
    gts[0]["wlh"][1::3, :2] = boxes.LARGE[:2] * 0.95
    cls_b, reg_b = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    for b, g in enumerate(gts):
        ref_c, ref_r, _ = _oracle_targets(oracle, anchors, g, 600)
        _check(cls_b[b], reg_b[b], ref_c, ref_r)


def test_batch_all_empty_max_batch_and_bad_arguments(gpu, oracle):
    import ctypes
    import torch
    from pp_amd import _lib, boxes, synth
    from pp_amd.targets import TargetAssigner, _vp
    cfg = boxes.AnchorConfig(40, 40)
    anchors = boxes.make_anchors(cfg)
    ta = TargetAssigner(cfg, canvas_height=80, device=gpu)
    empty = {"centers": np.zeros((0, 3)), "wlh": np.zeros((0, 3)), "yaw": np.zeros(0), "classes": np.zeros(0, np.int32)}
    c, r = ta.assign_batch([empty, empty, empty], check=True)
    assert c.shape == (3, ta.A, 9) and not c.any() and not r.any()
    # PP_MAX_BATCH samples, a different number of boxes each (incl. none), crowded small canvas
    gts = _ragged_batch(80, range(100, 132), [(7 * k) % 23 for k in range(32)])
    cls_b, reg_b = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    for b in (0, 1, 5, 17, 31):
        g = gts[b]
        if len(g["yaw"]) == 0:
            assert not cls_b[b].any() and not reg_b[b].any()
            continue
        ref_c, ref_r, _ = _oracle_targets(oracle, anchors, g, 80)
        _check(cls_b[b], reg_b[b], ref_c, ref_r)
    with pytest.raises(ValueError):
        ta.assign_batch(gts + [empty])                              # 33 samples
    # C ABI: a negative count is refused before anything is launched
    counts, packed = ta.upload_batch(gts[:2])
    bad = (ctypes.c_int32 * 2)(3, -1)
    stream = ctypes.c_void_p(torch.cuda.current_stream(ta.device).cuda_stream)
    out = torch.empty((2, ta.A, 9), dtype=torch.float32, device=gpu)
    rc = _lib.lib().pp_assign_targets_grid_batch_dev(
        ta._ctx.handle, stream, 2, bad, cfg.fm_height, cfg.fm_width, float(cfg.fm_scale), cfg.per_cell,
        _vp(ta.types), *[ctypes.c_void_p(packed.data_ptr())] * 6, ctypes.byref(ta._prm), _vp(out), _vp(out))
    assert rc == _lib.PP_ERR_VALUE and b"sample 1" in _lib.lib().pp_last_error()


def test_batch_many_boxes_tail_beyond_lds(gpu, oracle):
    """A sample with more ground truths than the tail keeps in LDS (> 2048: its column words live in the
    sample's own slice of the global scratch) next to small samples."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(5)
    cfg = boxes.AnchorConfig(40, 40)
    anchors = boxes.make_anchors(cfg)
    H = 80
    gts = [_crowded_gt(rng, 30, H, 10, np.array([30.0, 40.0])), _crowded_gt(rng, 2100, H, 64, np.array([36.0, 48.0])),
           _crowded_gt(rng, 12, H, 6, np.array([50.0, 20.0])), _crowded_gt(rng, 2060, H, 30, np.array([20.0, 60.0]))]
    ta = TargetAssigner(cfg, canvas_height=H, pos_thresh=0.45, device=gpu)
    for _ in range(2):
        cls_b, reg_b = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    for b, g in enumerate(gts):
        c_img, k_img = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], H)
        ref_c, ref_r, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                               anchors["yaw"], g["centers"], g["wlh"], g["yaw"], g["classes"], H,
                                               pos_thresh=0.45)
        _check(cls_b[b], reg_b[b], ref_c, ref_r)


def _soak_batches(ta, batches, refs, gpu, calls):
    """The batch form of _soak_single: call k runs batch order[k] into ONE pair of output tensors (the next call
    overwrites the rows this one wrote -- forced rows of the previous batch included)."""
    import torch
    from util import SideTraffic, check_targets, soak_order
    up = [ta.upload_batch(b) for b in batches]
    order = soak_order(len(batches), calls)
    first, bad = {}, torch.zeros((), dtype=torch.int64, device=gpu)
    out = None
    traffic = SideTraffic(gpu)
    for it, k in enumerate(order):
        if it % 3 == 0:
            traffic.flush()
        if it % 2 == 0:
            traffic.poke()
        c, r = ta.assign_batch_device(*up[k], out=out)
        out = (c, r)
        if k not in first:
            first[k] = (c.clone(), r.clone())
        else:
            bad += (torch.ne(c, first[k][0]).any() | torch.ne(r, first[k][1]).any()).to(torch.int64)
    traffic.close()
    torch.cuda.synchronize()
    assert ta._L.pp_iou_check(ta._ctx.handle, _stream(gpu)) == 0
    for k, (c, r) in first.items():
        for b in range(len(batches[k])):
            check_targets(c[b], r[b], *refs[k][b])
    assert int(bad.item()) == 0, f"{int(bad.item())} of {len(order)} calls differ from their own batch's result"
    return len(order)


def test_batch_repeated_calls_are_identical(gpu, oracle):
    """The alternating-input soak for the batch form: four tails run in one launch, each behind its own ticket, and
    consecutive calls carry DIFFERENT batches (different box counts per sample slot, an empty sample that moves from
    slot to slot) into the same output tensors -- a tail that read the previous call's list, or a zero fill that
    landed after a forced row, shows as a difference from the batch's own oracle result."""
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from util import oracle_targets_for
    for cfg, H in ((boxes.AnchorConfig(250, 250), 500), (boxes.AnchorConfig.reference_default(), 600)):
        anchors = boxes.make_anchors(cfg)
        batches = [_ragged_batch(H, (13, 14, 15, 16), (40, 35, 0, 22)), _ragged_batch(H, (21, 22, 23, 24), (12, 0, 40, 31)),
                   _ragged_batch(H, (31, 32, 33, 34), (0, 3, 40, 40)), _ragged_batch(H, (41, 42, 43, 44), (29, 40, 7, 0))]
        refs = [[oracle_targets_for(oracle, anchors, g, H, 0.6) for g in b] for b in batches]
        assert int((refs[0][0][1][:, 0] == 1).sum()) >= 20 and not refs[0][2][1].any()
        for src in (cfg, anchors):
            ta = TargetAssigner(src, canvas_height=H, device=gpu)
            assert _soak_batches(ta, batches, refs, gpu, 200) >= 200


def test_alternating_tail_kinds_on_one_context(gpu, oracle):
    """The tails differ in the scratch they touch and re-arm (box-centric fast tail: registers and LDS only; more
    than 256 pairs above the threshold or more than 64 boxes: the positives' hash table / per-anchor words in global
    memory, column words in global memory beyond the LDS tail).  Alternate sets that take DIFFERENT tails on one
    context: each call against its own oracle result."""
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from util import oracle_targets_for
    rng = np.random.default_rng(11)
    cfg, H = boxes.AnchorConfig(40, 40), 80
    anchors = boxes.make_anchors(cfg)
    sets = [_crowded_gt(rng, 30, H, 10, np.array([30.0, 40.0])), _crowded_gt(rng, 2100, H, 64, np.array([36.0, 48.0])),
            _crowded_gt(rng, 12, H, 6, np.array([50.0, 20.0])), _crowded_gt(rng, 300, H, 200, np.array([20.0, 60.0])),
            _crowded_gt(rng, 70, H, 30, np.array([44.0, 30.0]))]
    refs = [oracle_targets_for(oracle, anchors, s, H, 0.45) for s in sets]
    for src in (cfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=0.45, device=gpu)
        assert _soak_single(ta, sets, refs, gpu, 60) >= 60


def test_feature_map_scale_not_a_power_of_two(gpu, oracle):
    """The anchor grid's cell centres are (x + .5) / fm_scale (box_utils.py:137-138).  For a power-of-two scale the
    kernels multiply by the exact reciprocal; any other scale takes the true f64 division -- both must give the bits
    of the uploaded make_anchors arrays and the oracle's targets."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    for scale, H in ((0.4, 150), (0.25, 240), (1.0, 60), (0.3, 200)):
        cfg = boxes.AnchorConfig(60, 60, scale)
        anchors = boxes.make_anchors(cfg)
        gt = synth.gt_boxes(14, H, 21, margin=20.0)
        ref_c, ref_r, _ = _oracle_targets(oracle, anchors, gt, H)
        outs = []
        for src in (cfg, anchors):
            ta = TargetAssigner(src, canvas_height=H, device=gpu)
            cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
            cb, rb = ta.assign_batch([gt, gt], check=True)
            torch.cuda.synchronize()
            assert torch.equal(cb[0], cls_t) and torch.equal(cb[1], cls_t) and torch.equal(rb[1], reg_t)
            _check(cls_t, reg_t, ref_c, ref_r)
            outs.append((cls_t, reg_t))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), scale
        assert (ref_r[:, 0] == 1).sum() > 0


# --------------------------------------------------------------------------- the box-centric kernel's own corners
def _both_forms_and_oracle(gpu, oracle, cfg, H, gts, thresh, classes=9):
    """anchors on the fly (k_targets_gt: box-centric) and uploaded arrays (k_targets: anchor-centric) on the same
    batch: bit-equal to each other, and every sample against the oracle"""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    anchors = boxes.make_anchors(cfg)
    outs = []
    for src in (cfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=thresh, num_classes=classes, device=gpu)
        for _ in range(2):
            c, r = ta.assign_batch(gts, check=True)
        torch.cuda.synchronize()
        outs.append((c.clone(), r.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "the two forms differ"
    n_pos = []
    for b, g in enumerate(gts):
        if len(g["yaw"]) == 0:
            assert not outs[0][0][b].any() and not outs[0][1][b].any()
            continue
        c_img, k_img = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], H)
        ref_c, ref_r, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                               anchors["yaw"], g["centers"], g["wlh"], g["yaw"], g["classes"], H,
                                               pos_thresh=thresh, num_classes=classes)
        _check(outs[0][0][b], outs[0][1][b], ref_c, ref_r)
        n_pos.append(int((ref_r[:, 0] == 1).sum()))
    return n_pos


@pytest.mark.parametrize("thresh,lo,hi", [(0.45, 257, 900),       # more pairs above the threshold than the one-pass tail takes (566)
                                          (0.02, 1025, 10 ** 9)])  # ... and than the LDS table holds (7945): per-anchor words
def test_box_centric_many_positives(gpu, oracle, thresh, lo, hi):
    from pp_amd import boxes, synth
    cfg = boxes.AnchorConfig(120, 120)
    gts = [synth.gt_boxes(40, 240, 31, margin=30.0), synth.gt_boxes(12, 240, 32, margin=30.0)]
    n_pos = _both_forms_and_oracle(gpu, oracle, cfg, 240, gts, thresh)
    assert lo <= n_pos[0] <= hi, n_pos                  # the sample really takes the path this case is about


def test_box_centric_more_candidates_than_64_workgroups_hold(gpu, oracle):
    """at fm_scale = 1 with six anchors per cell a box has 23 x 23 x 6 = 3174 candidate anchors -- more than
    the 64 PAIR workgroups per box x 32 candidates the launch is clamped to, so every workgroup strides over several
    windows and, with a low threshold, one box has more pairs above it than 64 x 32.  The positive list is sized from
    the candidate count (nothing dropped, no error flag), and a second assigner with another grid on the SAME context
    kind re-arms its scratch (the layout key is compared field by field)."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    ref = boxes.AnchorConfig.reference_default()
    cfg = boxes.AnchorConfig(80, 80, 1.0, ref.dims, ref.yaws_deg, ref.zs)
    gts = [synth.gt_boxes(3, 80, 41, margin=25.0), synth.gt_boxes(2, 80, 42, margin=25.0)]
    n_pos = _both_forms_and_oracle(gpu, oracle, cfg, 80, gts, 0.05, classes=9)
    assert max(n_pos) > 700, n_pos
    # the same assigner object (one context) asked for another grid shape in between: results stay right
    ta = TargetAssigner(cfg, canvas_height=80, pos_thresh=0.05, device=gpu)
    c0, r0 = ta.assign_batch(gts, check=True)
    small = boxes.AnchorConfig(40, 40)
    tb = TargetAssigner(small, canvas_height=80, device=gpu)
    tb._ctx = ta._ctx                                   # two grids, ONE context
    g2 = synth.gt_boxes(6, 80, 5, margin=15.0)
    want = TargetAssigner(small, canvas_height=80, device=gpu).assign(g2["centers"], g2["wlh"], g2["yaw"], g2["classes"])
    got = tb.assign(g2["centers"], g2["wlh"], g2["yaw"], g2["classes"], check=True)
    c1, r1 = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(c0, c1) and torch.equal(r0, r1)


def test_more_than_eight_anchor_types_per_cell_take_the_anchor_centric_kernel(gpu, oracle):
    """The box-centric kernel keeps the anchor type table (and the per-type row constants) in LDS: eight types.  Nine per
    cell on the fly are routed through the anchor-centric k_targets by the host -- same results as the uploaded arrays
    and the oracle, next to an eight-type grid on the same context."""
    rng = np.random.default_rng(91)
    from pp_amd import boxes, synth
    for per_cell in (8, 9):
        dims = tuple((float(rng.uniform(6, 12)), float(rng.uniform(14, 26)), float(rng.uniform(1, 2))) for _ in range(per_cell))
        yaws = tuple(float(y) for y in rng.choice([0.0, 90.0, 45.0], per_cell))
        zs = tuple(float(z) for z in rng.uniform(0.4, 1.1, per_cell))
        cfg = boxes.AnchorConfig(30, 30, 0.5, dims, yaws, zs)
        gts = [synth.gt_boxes(9, 60, 51, margin=12.0), synth.gt_boxes(0, 60, 52), synth.gt_boxes(4, 60, 53, margin=12.0)]
        n_pos = _both_forms_and_oracle(gpu, oracle, cfg, 60, gts, 0.5)
        assert sum(n_pos) > 0


def test_box_centric_odd_sizes_misaligned_samples_and_boxes_off_the_map(gpu, oracle):
    """A = 81 anchors x 3 classes: the second and third sample's rows start off 16 bytes (the zero fill's head / tail
    paths); more list slots per box than the anchor-centric form has workgroups; boxes whose gate window lies
    partly or wholly outside the feature map; a sample without boxes in the middle."""
    from pp_amd import boxes
    rng = np.random.default_rng(77)
    cfg = boxes.AnchorConfig(9, 9, 0.5, ((5.0, 9.0, 1.5),), (30.0,), (0.6,))
    H = 18

    def gt(n, lo, hi):
        return {"centers": np.column_stack([rng.uniform(lo, hi, n), rng.uniform(lo, hi, n), rng.uniform(0, 1, n)]),
                "wlh": np.column_stack([rng.uniform(4, 7, n), rng.uniform(7, 11, n), rng.uniform(1, 2, n)]),
                "yaw": rng.uniform(-np.pi, np.pi, n), "classes": rng.integers(0, 3, n).astype(np.int32)}
    gts = [gt(5, 2, 16), gt(0, 0, 1), gt(7, -30, 48), gt(3, 1000.0, 2000.0), gt(6, 0, 18)]
    _both_forms_and_oracle(gpu, oracle, cfg, H, gts, 0.3, classes=3)


def test_box_centric_centre_not_finite(gpu, oracle):
    """A box whose image-space centre is NaN passes the reference's centre gate with EVERY anchor (NaN compares
    false, pillars.cpp:418-419): the box-centric kernel walks the whole map for it; with finite corners it has real
    overlaps.  The two kernel forms must agree (the oracle's numpy path is not asked: its argmax meets NaN)."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    cfg = boxes.AnchorConfig(40, 40)
    anchors = boxes.make_anchors(cfg)
    g = synth.gt_boxes(6, 80, 5, margin=15.0)
    outs = []
    for src in (cfg, anchors):
        ta = TargetAssigner(src, canvas_height=80, device=gpu)
        counts, packed = ta.upload_batch([g, g])
        T = sum(counts)
        packed[T * 8 + 3 * 2] = float("nan")          # centres_img[2].x of sample 0 (its corners stay finite)
        c, r = ta.assign_batch_device(counts, packed)
        torch.cuda.synchronize()
        outs.append((c.clone(), r.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0][1], TargetAssigner(cfg, canvas_height=80, device=gpu).assign(
        g["centers"], g["wlh"], g["yaw"], g["classes"])[0])
    assert (outs[0][1][0][:, 0] == 1).sum().item() > 0


def test_box_centric_falls_back_beyond_its_grid_limit(gpu, oracle):
    """Anchors on the fly with more boxes than the box-centric launch can hold in one grid dimension (64 zero-fill
    workgroups + boxes x workgroups per box <= 65535: about eleven thousand boxes at six per box) go through the
    anchor-centric kernel instead of being refused; same results, next to an ordinary sample in the same batch."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(77)
    cfg = boxes.AnchorConfig(40, 40)
    anchors = boxes.make_anchors(cfg)
    H = 80
    gts = [_crowded_gt(rng, 11500, H, 64, np.array([36.0, 48.0])), _crowded_gt(rng, 25, H, 8, np.array([30.0, 40.0]))]
    ta = TargetAssigner(cfg, canvas_height=H, pos_thresh=0.45, device=gpu)
    for _ in range(2):
        cls_b, reg_b = ta.assign_batch(gts, check=True)
    torch.cuda.synchronize()
    for b, g in enumerate(gts):
        c_img, k_img = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], H)
        ref_c, ref_r, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                               anchors["yaw"], g["centers"], g["wlh"], g["yaw"], g["classes"], H,
                                               pos_thresh=0.45)
        _check(cls_b[b], reg_b[b], ref_c, ref_r)
    # ... and the small sample alone afterwards (box-centric again: the scratch is re-armed for the other form)
    cls_1, reg_1 = ta.assign_batch(gts[1:], check=True)
    assert torch.equal(cls_1[0], cls_b[1]) and torch.equal(reg_1[0], reg_b[1])


def test_positive_list_overflow_is_reported_and_the_next_call_is_right(gpu, oracle):
    """The positive list of the box-centric kernel holds one entry per candidate anchor of a box's +-10 centre window
    (pillars.cpp:418-419).  A box whose image-space centre is not finite on ONE axis passes that gate with a whole
    band of the map (NaN compares false): with a box large enough and a threshold low enough, more pairs exceed the
    threshold than the list holds.  The launch must say so (error bit 2 -> ValueError, not a winding error, entries
    dropped rather than written out of bounds), and the NEXT call on the same context must be right again."""
    import torch
    from pp_amd import boxes, synth
    from pp_amd.targets import TargetAssigner
    from util import check_targets, oracle_targets_for
    cfg, H = boxes.AnchorConfig(64, 64), 128
    anchors = boxes.make_anchors(cfg)
    ta = TargetAssigner(cfg, canvas_height=H, pos_thresh=0.001, device=gpu)
    big = {"centers": np.array([[64.0, 64.0, 0.5]]), "wlh": np.array([[100.0, 100.0, 2.0]]),
           "yaw": np.array([0.0]), "classes": np.array([3], np.int32)}
    g = list(ta._gt_to_device(big["centers"], big["wlh"], big["yaw"], big["classes"]))
    # finite centre: at most 13 x 13 x 2 candidates, the list holds them all
    c0, r0 = ta.assign_device(*g, check=True)
    check_targets(c0, r0, *oracle_targets_for(oracle, anchors, big, H, 0.001))
    n_pos = int((r0[:, 0] == 1).sum())
    assert 100 < n_pos <= 13 * 13 * 2
    g[1] = g[1].clone()
    g[1][0, 0] = float("nan")                       # the centre's x is not finite: every column passes the gate
    with pytest.raises(ValueError, match="more pairs above the threshold"):
        ta.assign_device(*g, check=True)
    # the context is usable again, and right: the same box with its finite centre, then an ordinary sample
    g2 = ta._gt_to_device(big["centers"], big["wlh"], big["yaw"], big["classes"])
    c1, r1 = ta.assign_device(*g2, check=True)
    assert torch.equal(c1, c0) and torch.equal(r1, r0)
    gt = synth.gt_boxes(12, H, 3, margin=20.0)
    tb = TargetAssigner(cfg, canvas_height=H, device=gpu)
    c2, r2 = tb.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
    check_targets(c2, r2, *oracle_targets_for(oracle, anchors, gt, H, 0.6))
    # a wrong winding AND an overflow in one launch: the overflow is what is reported (the scratch must be re-armed),
    # and the call after that is clean
    g3 = [t.clone() for t in g]
    g3[0] = g3[0].flip(1)                           # corners in the opposite order
    with pytest.raises(ValueError):
        ta.assign_device(*g3, check=True)
    c3, r3 = ta.assign_device(*g2, check=True)
    assert torch.equal(c3, c0) and torch.equal(r3, r0)
