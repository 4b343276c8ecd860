"""Reference-run vectors of create_target / make_target (tests/golden/targets_ref_golden.npz, made by
tests/golden/make_targets_ref_golden.py by importing /root/reference/utils/box_utils.py:162-232, 70-109 in the
build container and CALLING the reference's own functions).

What these pin: all of create_target's post-IoU logic (strict threshold, first argmax, the np.nonzero filter that
drops a best anchor 0, one-hot zero-and-reset of forced rows, duplicates, write order) and all of make_target's
arithmetic.  What they do not pin: the IoU values (the generator binds ``data.pillars.make_ious`` to the oracle's --
the Boost build is not obtainable here), ``Box.bottom_corners`` and ``Quaternion.yaw_pitch_roll`` (inputs there).

CPU: the oracle's restatement must equal the reference's outputs (bit-exact; same numpy, same statements).
GPU: the HIP target assignment (uploaded anchor arrays AND on-the-fly anchor grid) must equal them: classes,
positive flags and orientation bits exact, regression values within 1e-6 (device log / sin vs glibc, f32)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "targets_ref_golden.npz")
REG_TOL = 1e-6


@pytest.fixture(scope="module")
def cases():
    g = dict(np.load(GOLD))
    out = {}
    for name in g["__cases__"]:
        out[str(name)] = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(str(name) + "/")}
    assert {"hotpath", "c3_small", "default_anchor_set", "best_anchor_is_0", "duplicate_forced",
            "exactly_at_threshold", "just_below_threshold", "ties", "yaw_quadrants", "no_overlap"} <= set(out)
    return out


def _setup(c):
    from pp_amd import boxes
    fm = c["fm"]
    acfg = boxes.AnchorConfig(int(fm[0]), int(fm[1]), float(fm[2]), tuple(tuple(d) for d in c["dims"]),
                              tuple(c["yaws_deg"]), tuple(c["zs"]))
    gt = {"centers": c["gt_centers"], "wlh": c["gt_wlh"], "yaw": c["gt_yaw"], "classes": c["gt_classes"]}
    return acfg, boxes.make_anchors(acfg), gt, int(fm[3]), float(fm[4])


def test_oracle_equals_the_reference_run(oracle, cases):
    from pp_amd import boxes
    for name, c in cases.items():
        if "cls" not in c:
            continue
        acfg, anchors, gt, H, thresh = _setup(c)
        c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
        cls_t, reg_t, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                               anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                                               pos_thresh=thresh)
        assert cls_t.shape == c["cls"].shape and reg_t.shape == c["reg"].shape, name
        assert np.array_equal(cls_t, c["cls"]), name
        assert np.array_equal(reg_t, c["reg"]), name          # same statements under the same numpy: bit-equal


def test_anchor_grid_equals_the_reference_run(cases):
    """make_anchor_boxes executed from the reference's source (order (y, x, d), centres, the anchor_xy corner
    choice) == boxes.make_anchors and the oracle's loop restatement."""
    from oracle import oracle as O
    from pp_amd import boxes
    n = 0
    for name, c in cases.items():
        if "anchor_corners" not in c:
            continue
        fm = c["fm"]
        acfg = boxes.AnchorConfig(int(fm[0]), int(fm[1]), float(fm[2]), tuple(tuple(d) for d in c["dims"]),
                                  tuple(c["yaws_deg"]), tuple(c["zs"]))
        a = boxes.make_anchors(acfg)
        assert np.array_equal(a["corners"], c["anchor_corners"]) and np.array_equal(a["centers"], c["anchor_centers"])
        assert np.array_equal(a["xy"], c["anchor_xy"])
        oc, on, _, _ = O.make_anchor_boxes(acfg.fm_height, acfg.fm_width, acfg.fm_scale, acfg.dims, acfg.yaws_deg, acfg.zs)
        assert np.allclose(oc, c["anchor_corners"], rtol=0, atol=1e-12) and np.array_equal(on, c["anchor_centers"])
        assert np.allclose(O.anchor_xy_rows(oc, acfg.yaws_deg), c["anchor_xy"], rtol=0, atol=1e-12)
        n += 1
    assert n == 2


def test_boxes_to_image_space_equals_the_reference_run(cases):
    """boxes_to_image_space (box_utils.py:19-32) executed from the reference's source on field-holder boxes: the
    y flip of centres and corners and the [3,4] -> [4,2] corner transpose == boxes.boxes_to_image_space (what the
    device path uploads) and the oracle's restatement."""
    from oracle import oracle as O
    from pp_amd import boxes
    n = 0
    for name, c in cases.items():
        if "centers_img" not in c:
            continue
        H = float(c["fm"][3])
        mc, mk = boxes.boxes_to_image_space(c["gt_centers"], c["gt_wlh"], c["gt_yaw"], H)
        assert mc.shape == c["centers_img"].shape and mk.shape == c["corners_img"].shape == (len(mc), 4, 2), name
        assert np.array_equal(mc, c["centers_img"]) and np.array_equal(mk, c["corners_img"]), name
        oc, ok = O.boxes_to_image_space(c["gt_centers"], c["gt_wlh"], c["gt_yaw"], H)
        assert np.array_equal(oc, c["centers_img"]), name
        assert np.allclose(ok, c["corners_img"], rtol=0, atol=1e-12), name     # the oracle's own cos / sin loop
        assert np.array_equal(c["centers_img"][:, 1], (H - 1) - c["gt_centers"][:, 1]), name
        n += 1
    assert n == 3


def test_the_vectors_cover_the_quirks(cases):
    """the fixture is only worth its name if the branches are in it"""
    c = cases["duplicate_forced"]
    assert (c["cls"].sum(1) == 2).sum() == 1                   # one anchor carries two classes
    c = cases["best_anchor_is_0"]
    assert not c["cls"][0].any() and c["reg"][:, 0].sum() == 1   # the box on anchor 0 left nothing behind
    a, b = cases["exactly_at_threshold"], cases["just_below_threshold"]
    assert a["reg"][:, 0].sum() + 1 == b["reg"][:, 0].sum()    # strict >: one anchor flips with one ulp
    y = cases["yaw_quadrants"]["reg"]
    assert {0.0, 1.0} == set(np.unique(y[y[:, 0] == 1][:, 8]))
    assert not cases["no_overlap"]["cls"].any()


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["arrays", "grid"])
def test_hip_target_assignment_equals_the_reference_run(gpu, cases, source):
    import torch
    from pp_amd.targets import TargetAssigner
    for name, c in cases.items():
        if "cls" not in c:
            continue
        acfg, anchors, gt, H, thresh = _setup(c)
        ta = TargetAssigner(anchors if source == "arrays" else acfg, canvas_height=H, pos_thresh=thresh, device=gpu)
        cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
        torch.cuda.synchronize()
        cls_t, reg_t = cls_t.cpu().numpy(), reg_t.cpu().numpy()
        ref_c, ref_r = c["cls"].astype(np.float32), c["reg"].astype(np.float32)
        assert np.array_equal(cls_t, ref_c), name
        assert np.array_equal(reg_t[:, 0], ref_r[:, 0]) and np.array_equal(reg_t[:, 8], ref_r[:, 8]), name
        assert np.abs(reg_t - ref_r).max() <= REG_TOL, name


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["arrays", "grid"])
def test_hip_batched_target_assignment_equals_the_reference_run(gpu, cases, source):
    """The batch form (one launch for the samples of a step): the fixture's cases that share an anchor set go
    through ONE pp_assign_targets*_batch_dev call each and must equal the reference's own outputs."""
    import torch
    from pp_amd.targets import TargetAssigner
    groups = {}
    for name, c in cases.items():
        if "cls" not in c:
            continue
        key = (c["fm"].tobytes(), c["dims"].tobytes(), c["yaws_deg"].tobytes(), c["zs"].tobytes())
        groups.setdefault(key, []).append(name)
    assert max(len(v) for v in groups.values()) >= 2           # at least one real batch
    for names in groups.values():
        acfg, anchors, _, H, thresh = _setup(cases[names[0]])
        ta = TargetAssigner(anchors if source == "arrays" else acfg, canvas_height=H, pos_thresh=thresh, device=gpu)
        gts = [_setup(cases[n])[2] for n in names]
        cls_b, reg_b = ta.assign_batch(gts, check=True)
        torch.cuda.synchronize()
        for b, name in enumerate(names):
            c = cases[name]
            cls_t, reg_t = cls_b[b].cpu().numpy(), reg_b[b].cpu().numpy()
            ref_c, ref_r = c["cls"].astype(np.float32), c["reg"].astype(np.float32)
            assert np.array_equal(cls_t, ref_c), name
            assert np.array_equal(reg_t[:, 0], ref_r[:, 0]) and np.array_equal(reg_t[:, 8], ref_r[:, 8]), name
            assert np.abs(reg_t - ref_r).max() <= REG_TOL, name


@pytest.mark.gpu
def test_uploaded_ground_truths_equal_the_reference_run(gpu, cases):
    """What TargetAssigner puts on the DEVICE for a batch (upload_batch: image-space corners and centres next to the
    canvas-space box fields) against boxes_to_image_space run from the reference's source."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    names = [n for n, c in cases.items() if "centers_img" in c]
    for name in names:
        c = cases[name]
        H = int(c["fm"][3])
        ta = TargetAssigner(boxes.AnchorConfig(8, 8), canvas_height=H, device=gpu)
        g = {"centers": c["gt_centers"], "wlh": c["gt_wlh"], "yaw": c["gt_yaw"],
             "classes": np.zeros(len(c["gt_yaw"]), np.int32)}
        counts, packed = ta.upload_batch([g, g])
        T = sum(counts)
        host = packed.cpu().numpy()
        corners, cimg = host[:T * 8].reshape(T, 4, 2), host[T * 8:T * 11].reshape(T, 3)
        n = counts[0]
        for o in (0, n):
            assert np.array_equal(corners[o:o + n], c["corners_img"]), name
            assert np.array_equal(cimg[o:o + n], c["centers_img"]), name
