"""Generates tests/golden/targets_ref_golden.npz by running the REFERENCE's own
``create_target`` / ``make_target`` (/root/reference/utils/box_utils.py:162-232, 70-109) in this
container (container-only; the reference never travels to the GPU box, the fixture does).

utils/box_utils.py imports packages that are absent from the image and not installable.  None of
them carries arithmetic on the path that is run here, so the import is satisfied with
arithmetic-free stand-ins:

* ``easydict``                      an attribute dict (the one make_model_golden.py uses);
* ``pyquaternion``, ``lyft_dataset_sdk.*``   modules whose names (Quaternion, Box, LidarPointCloud,
                                    transform_matrix, LyftDataset, points_in_box) are bound to a class
                                    that raises when touched -- create_target / make_target never
                                    construct or call them, they only READ attributes of the box
                                    objects handed in;
* ``data.pillars``                  ``make_ious`` bound to the oracle's (oracle/pp_oracle.c): the
                                    compiled reference module cannot be built here (Boost absent).
* the box objects                   plain attribute holders with exactly the attributes the two
                                    functions read: ``.center``, ``.wlh``, ``.name``,
                                    ``.orientation.yaw_pitch_roll`` (a tuple whose [0] is the yaw).

WHAT THIS PINS: every statement of create_target after the ``make_ious`` call (max / argmax /
strict threshold / the np.nonzero filter that drops a best anchor 0 / class one-hots with the
zero-and-reset of the forced rows / duplicate forced anchors / write order of the regression rows)
and every arithmetic statement of make_target (diagonal, y flip, quotients, logs, yaw folding, sine,
orientation bit) -- executed by the reference's own source text under this image's numpy.
Also run from the reference's source: make_anchor_boxes (box_utils.py:111-159) on two small maps -- the anchor
ORDER (y, x, d), the centre formula and the anchor_xy corner choice; the corner values come from this repo's
bottom_corners_xy standing in for the absent SDK method (section 5 below) -- and boxes_to_image_space
(box_utils.py:19-32) on three box sets: the y flip of centres and corners and the corner transpose (section 6).
WHAT IT DOES NOT PIN: the IoU values themselves (they come from the oracle: Boost.Geometry stays
unpinned), ``Box.bottom_corners`` (corner arrays are inputs here, made by the repo's boxes.py) and
``Quaternion.yaw_pitch_roll`` (the yaw is handed over as a number; the SDK would derive it from a
quaternion built with ``degrees=90`` -- a 1-ulp difference there cannot be checked in this image).

Run:  python tests/golden/make_targets_ref_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "targets_ref_golden.npz")

tmp = tempfile.mkdtemp()
with open(os.path.join(tmp, "easydict.py"), "w") as f:
    f.write("class EasyDict(dict):\n"
            "    def __getattr__(self, k):\n"
            "        try:\n            return self[k]\n"
            "        except KeyError:\n            raise AttributeError(k)\n"
            "    def __setattr__(self, k, v):\n        self[k] = v\n")
sys.path.insert(0, tmp)
sys.path.insert(0, ROOT)
import pp_amd  # noqa: E402,F401
from oracle import oracle as O  # noqa: E402
from pp_amd import boxes, synth  # noqa: E402

O.build()


class _Absent:
    """stands for a name of an absent third-party package; any use is an error"""
    def __init__(self, *a, **k):
        raise RuntimeError("absent third-party class used: this generator must not need it")


def _stub(name, **names):
    m = types.ModuleType(name)
    m.__dict__.update(names)
    sys.modules[name] = m
    return m


_stub("pyquaternion", Quaternion=_Absent)
_stub("lyft_dataset_sdk")
_stub("lyft_dataset_sdk.utils")
_stub("lyft_dataset_sdk.utils.data_classes", LidarPointCloud=_Absent, Box=_Absent)
_stub("lyft_dataset_sdk.utils.geometry_utils", transform_matrix=_Absent, points_in_box=_Absent)
_stub("lyft_dataset_sdk.lyftdataset", LyftDataset=_Absent)
_pillars = types.ModuleType("data.pillars")
_pillars.make_ious = O.make_ious                      # the oracle's IoU (see the header)
sys.modules["data.pillars"] = _pillars

sys.path.insert(0, REF)
from config import cfg  # noqa: E402
import data as _ref_data  # noqa: E402
_ref_data.pillars = _pillars
import utils.box_utils as ref  # noqa: E402   the reference's own source, unmodified

assert ref.create_target.__code__.co_filename.startswith(REF)


class _Orientation:
    def __init__(self, yaw):
        self.yaw_pitch_roll = (float(yaw), 0.0, 0.0)


class _BoxFields:
    """the attributes create_target / make_target read from a lyft Box"""
    def __init__(self, center, wlh, yaw, name=None):
        self.center = np.array(center, np.float64)
        self.wlh = np.array(wlh, np.float64)
        self.orientation = _Orientation(yaw)
        self.name = name


NAMES = list(cfg.DATA.CLASS_NAMES)
cfg.DATA.NAME_TO_IND = getattr(cfg.DATA, "NAME_TO_IND", None) or {n: i for i, n in enumerate(NAMES)}
assert [cfg.DATA.NAME_TO_IND[n] for n in NAMES] == list(range(9))


def run_reference(anchors, gt, canvas_height, pos_thresh):
    """anchors: dict of boxes.make_anchors; gt: dict centers/wlh/yaw/classes (canvas space)"""
    cfg.DATA.CANVAS_HEIGHT = canvas_height        # read at call time (box_utils.py:29-30,83)
    cfg.DATA.IOU_POS_THRESH = pos_thresh          # read at call time (box_utils.py:178)
    a_list = [_BoxFields(c, w, y) for c, w, y in zip(anchors["centers"], anchors["wlh"], anchors["yaw"])]
    g_list = [_BoxFields(c, w, y, NAMES[int(k)]) for c, w, y, k in
              zip(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])]
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], canvas_height)
    cls_t, reg_t = ref.create_target(anchors["corners"], k_img, anchors["centers"], c_img, a_list, g_list)
    return cls_t, reg_t, c_img, k_img


def gt_dict(rows):
    """rows of (x, y, z, w, l, h, yaw, class)"""
    r = np.asarray(rows, np.float64).reshape(-1, 8)
    return {"centers": r[:, 0:3].copy(), "wlh": r[:, 3:6].copy(), "yaw": r[:, 6].copy(),
            "classes": r[:, 7].astype(np.int32)}


cases = {}


def add(name, acfg, gt, H, thresh=0.6):
    anchors = boxes.make_anchors(acfg)
    cls_t, reg_t, c_img, k_img = run_reference(anchors, gt, H, thresh)
    cases[name] = dict(fm=np.array([acfg.fm_height, acfg.fm_width, acfg.fm_scale, H, thresh], np.float64),
                       dims=np.asarray(acfg.dims, np.float64), yaws_deg=np.asarray(acfg.yaws_deg, np.float64),
                       zs=np.asarray(acfg.zs, np.float64),
                       gt_centers=gt["centers"], gt_wlh=gt["wlh"], gt_yaw=gt["yaw"], gt_classes=gt["classes"],
                       cls=cls_t, reg=reg_t)
    pos = int((reg_t[:, 0] == 1).sum())
    multi = int((cls_t.sum(1) > 1).sum())
    print(f"{name}: A={len(cls_t)} G={len(gt['yaw'])} positives={pos} multi-class rows={multi}")
    return anchors, cls_t, reg_t


# 1. the hot-path fixture's target case (12x12x2 anchors, 7 boxes, an exact match and a duplicate)
gold = dict(np.load(os.path.join(HERE, "hotpath_golden.npz")))
add("hotpath", boxes.AnchorConfig(12, 12),
    {"centers": gold["tgt_gt_centers"], "wlh": gold["tgt_gt_wlh"], "yaw": gold["tgt_gt_yaw"],
     "classes": gold["tgt_gt_classes"]}, 24, 0.5)

# 2. BASELINE config 3's generator at a smaller map: 60x60x2 anchors, 24 boxes, yaw over (-pi, pi)
add("c3_small", boxes.AnchorConfig(60, 60), synth.gt_boxes(24, 120, 11, margin=14.0), 120)

# 3. the reference's shipped anchor set (6 per cell: small/med/large x 0/90 degrees, config.py:109-116),
#    40x40 map; boxes of every size group
dflt = boxes.AnchorConfig.reference_default()
d40 = boxes.AnchorConfig(40, 40, 0.5, dflt.dims, dflt.yaws_deg, dflt.zs)
rng = np.random.default_rng(2024)
rows = []
for i in range(18):
    grp = (boxes.SMALL, boxes.MED, boxes.LARGE)[i % 3]
    w, l, h = grp * rng.uniform(0.85, 1.2, 3)
    yaw = rng.uniform(-np.pi, np.pi) if i % 2 else rng.choice([0.0, np.pi / 2, -np.pi / 2]) + rng.normal(0, 0.05)
    rows.append((rng.uniform(12, 68), rng.uniform(12, 68), rng.uniform(0.2, 1.4), w, l, h, yaw, i % 9))
add("default_anchor_set", d40, gt_dict(rows), 80)

# 4. quirks, on a 6x6x2 map of SMALL non-overlapping anchors (1.5 x 1.8 cells, centres at 1, 3, 5, ... cells;
#    H = 12; a yaw-0 anchor spans +-0.9 in x and +-0.75 in y, its 90-degree twin the other way round)
QW, QL, QH = 1.5, 1.8, 1.5
q = boxes.AnchorConfig(6, 6, 0.5, ((QW, QL, QH), (QW, QL, QH)), (0.0, 90.0), (0.75, 0.75))
qa = boxes.make_anchors(q)
HQ = 12


def at(ix, iy, dx=0.0, dy=0.0, yaw=0.0, cls=0, w=QW, l=QL, h=QH, z=0.7):
    """a box whose IMAGE-space centre is anchor cell (ix, iy)'s centre + (dx, dy)"""
    return (2 * ix + 1 + dx, (HQ - 1) - (2 * iy + 1 + dy), z, w, l, h, yaw, cls)


def aidx(ix, iy, d):
    return (iy * 6 + ix) * 2 + d


# 4a. box 0's best anchor is index 0 (IoU 0.5 < 0.6): dropped by the np.nonzero filter although it IS the
#     column maximum; box 1 has the same overlap with anchor (2,1,0) and is forced; box 2 touches nothing
_, cls_t, reg_t = add("best_anchor_is_0", q, gt_dict([at(0, 0, dx=0.6, cls=4), at(2, 1, dx=0.6, cls=2),
                                                      at(40, 40, cls=1)]), HQ)
assert np.nonzero(reg_t[:, 0])[0].tolist() == [aidx(2, 1, 0)] and cls_t[aidx(2, 1, 0), 2] == 1 and cls_t.sum() == 1
# 4b. duplicates and overwrites: boxes 0 and 1 (classes 1, 6) both force anchor (2,1,0) from either side ->
#     both classes set, the LAST box's regression row stays; box 2 sits exactly on anchor (4,2,0) and also
#     makes its 90-degree twin (4,2,1) positive by threshold (IoU .714, class 3); box 3 (class 7, rotated)
#     has that twin as its best anchor below the threshold -> the twin's row is zeroed and re-set to class 7
#     and its regression row is rewritten from box 3
_, cls_t, reg_t = add("duplicate_forced", q, gt_dict([at(2, 1, dx=0.6, cls=1), at(2, 1, dx=-0.6, cls=6, z=0.9),
                                                      at(4, 2, cls=3), at(4, 2, dy=0.5, yaw=np.pi / 2, cls=7)]), HQ)
assert cls_t[aidx(2, 1, 0)].tolist() == [0, 1, 0, 0, 0, 0, 1, 0, 0]
assert cls_t[aidx(4, 2, 0)].tolist() == [0, 0, 0, 1, 0, 0, 0, 0, 0]
assert cls_t[aidx(4, 2, 1)].tolist() == [0, 0, 0, 0, 0, 0, 0, 1, 0]
assert reg_t[aidx(2, 1, 0), 1] < 0 and reg_t[aidx(4, 2, 1), 2] > 0      # rows of box 1 (dx<0) and box 3 (dy>0)
# 4c. an IoU exactly AT the threshold is not positive (strict >): the threshold is set to the twin's IoU
tie_gt = gt_dict([at(3, 3, dx=0.1, cls=3)])
c_img, k_img = boxes.boxes_to_image_space(tie_gt["centers"], tie_gt["wlh"], tie_gt["yaw"], HQ)
io = np.zeros((q.num_anchors, 1))
O.make_ious(qa["corners"], k_img, qa["centers"], c_img, io)
twin = float(io[aidx(3, 3, 1), 0])
assert 0 < twin < io[aidx(3, 3, 0), 0]
_, cls_t, _ = add("exactly_at_threshold", q, tie_gt, HQ, twin)
assert cls_t[aidx(3, 3, 0), 3] == 1 and not cls_t[aidx(3, 3, 1)].any()
_, cls_t, _ = add("just_below_threshold", q, tie_gt, HQ, float(np.nextafter(twin, 0.0)))
assert cls_t[aidx(3, 3, 0), 3] == 1 and cls_t[aidx(3, 3, 1), 3] == 1
# 4d. ties.  Boxes 0 and 1 are identical (classes 2, 5) and sit on anchor (1,4,0): the twin (1,4,1) is positive by
#     threshold for both with EQUAL IoU -> first argmax = box 0 = class 2.  Box 2 lies midway between anchors
#     (3,4,0) and (4,4,0), below the threshold with both: the column argmax takes the first of equal values
_, cls_t, reg_t = add("ties", q, gt_dict([at(1, 4, cls=2), at(1, 4, cls=5), at(3, 4, dx=1.0, cls=8)]), HQ)
assert cls_t[aidx(1, 4, 1)].tolist() == [0, 0, 1, 0, 0, 0, 0, 0, 0]
assert cls_t[aidx(1, 4, 0)].tolist() == [0, 0, 1, 0, 0, 1, 0, 0, 0]
assert int(cls_t[aidx(3, 4, 0), 8] + cls_t[aidx(4, 4, 0), 8]) == 1
# 4e. yaw in each quadrant and on the folding edges (make_target's branches, box_utils.py:92-102), against
#     anchors of both yaws
yaws = [0.3, 1.2, np.pi / 2, 2.0, 3.0, np.pi, -0.3, -1.2, -np.pi / 2, -2.0, -3.0, -np.pi, 1e-9, np.pi / 2 - 1e-9,
        np.nextafter(np.pi / 2, 0), np.nextafter(-np.pi / 2, 0)]
rows = [at(1 + i % 5, 1 + i // 5, dx=0.15, dy=-0.1, yaw=yw, cls=i % 9, w=QW * 1.05, l=QL * 0.95, h=QH * 1.1,
           z=0.6 + 0.05 * i) for i, yw in enumerate(yaws)]
_, cls_t, reg_t = add("yaw_quadrants", q, gt_dict(rows), HQ, 0.3)
assert set(np.unique(reg_t[:, 8])) == {0.0, 1.0}
# 4f. no box reaches any anchor
_, cls_t, reg_t = add("no_overlap", q, gt_dict([at(60, 60, yaw=0.2)]), HQ)
assert not cls_t.any() and not reg_t.any()

# 5. make_anchor_boxes (box_utils.py:111-159) run from the reference's source: the loop order (y, x, d), the centre
#    formula, the per-type size / yaw / z lookup and the anchor_xy corner choice (yaw > 0: corners 1,3, else 2,0).
#    The lyft Box / pyquaternion Quaternion it constructs are FIELD HOLDERS here; Box.bottom_corners() -- the
#    SDK's arithmetic, absent -- is answered by this repo's boxes.bottom_corners_xy (so the corner VALUES are
#    not pinned by this, their assembly into the four arrays is).
class _QuatFields:
    def __init__(self, axis=None, degrees=None, radians=None):
        assert list(axis) == [0, 0, 1] and radians is None
        self.degrees = degrees
        self.yaw_pitch_roll = (float(np.deg2rad(degrees)), 0.0, 0.0)     # what boxes.py uses (recalled, see DESIGN)


class _AnchorBoxFields:
    def __init__(self, center=None, size=None, orientation=None):
        self.center, self.wlh, self.orientation = np.array(center, np.float64), np.array(size, np.float64), orientation

    def bottom_corners(self):
        xy = boxes.bottom_corners_xy(self.center, self.wlh, np.float64(self.orientation.yaw_pitch_roll[0]))
        return np.vstack([xy.T, np.zeros((1, 4))])                       # [3,4] like the SDK; only xy is read


ref.Quaternion, ref.Box = _QuatFields, _AnchorBoxFields
for tag, acfg in (("anchors_c3_8x5", boxes.AnchorConfig(8, 5)),
                  ("anchors_default_6x7", boxes.AnchorConfig(6, 7, 0.5, dflt.dims, dflt.yaws_deg, dflt.zs))):
    cfg.DATA.FM_HEIGHT, cfg.DATA.FM_WIDTH, cfg.DATA.FM_SCALE = acfg.fm_height, acfg.fm_width, acfg.fm_scale
    cfg.DATA.ANCHOR_DIMS = [np.array(d) for d in acfg.dims]
    cfg.DATA.ANCHOR_YAWS, cfg.DATA.ANCHOR_ZS = list(acfg.yaws_deg), list(acfg.zs)
    b_list, corners, centers, xy = ref.make_anchor_boxes()               # reference-run
    mine = boxes.make_anchors(acfg)
    assert np.array_equal(corners, mine["corners"]) and np.array_equal(centers, mine["centers"])
    assert np.array_equal(xy, mine["xy"])
    assert np.array_equal(np.stack([b.wlh for b in b_list]), mine["wlh"])
    assert np.array_equal(np.array([b.orientation.yaw_pitch_roll[0] for b in b_list]), mine["yaw"])
    cases[tag] = dict(fm=np.array([acfg.fm_height, acfg.fm_width, acfg.fm_scale, 0, 0], np.float64),
                      dims=np.asarray(acfg.dims, np.float64), yaws_deg=np.asarray(acfg.yaws_deg, np.float64),
                      zs=np.asarray(acfg.zs, np.float64), anchor_corners=corners, anchor_centers=centers, anchor_xy=xy)
    print(f"{tag}: make_anchor_boxes from the reference's source == boxes.make_anchors ({len(corners)} anchors)")

# 6. boxes_to_image_space (box_utils.py:19-32) run from the reference's source on field-holder boxes: the y flip
#    (CANVAS_HEIGHT - 1) - y of centres and corners, the [3,4] -> [4,2] corner transpose, the stacking.  As in
#    section 5, Box.bottom_corners() -- the SDK's arithmetic -- is answered by boxes.bottom_corners_xy, so the
#    corner VALUES before the flip are this repo's; what is pinned is everything the reference's function does.
for tag, n_box, Hc, seed in (("image_space_c3", 40, 500, 0), ("image_space_default", 25, 600, 5),
                             ("image_space_one_box", 1, 41, 2)):
    g = synth.gt_boxes(n_box, Hc, seed, margin=min(50.0, Hc / 4))
    g["wlh"][::3] *= 0.6
    cfg.DATA.CANVAS_HEIGHT = Hc
    held = [_AnchorBoxFields(center=c, size=w, orientation=_Orientation(y))
            for c, w, y in zip(g["centers"], g["wlh"], g["yaw"])]
    before = [b.center.copy() for b in held]
    centers_img, corners_img = ref.boxes_to_image_space(held)              # reference-run
    assert ref.boxes_to_image_space.__code__.co_filename.startswith(REF)
    assert all(np.array_equal(b.center, c0) for b, c0 in zip(held, before))    # it copies the centres
    mine_c, mine_k = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], Hc)
    assert np.array_equal(mine_c, centers_img) and np.array_equal(mine_k, corners_img)
    cases[tag] = dict(fm=np.array([0, 0, 0, Hc, 0], np.float64), gt_centers=g["centers"], gt_wlh=g["wlh"],
                      gt_yaw=g["yaw"], centers_img=centers_img, corners_img=corners_img)
    print(f"{tag}: boxes_to_image_space from the reference's source == boxes.boxes_to_image_space ({n_box} boxes)")

flat = {}
for name, c in cases.items():
    for k, v in c.items():
        flat[f"{name}/{k}"] = v
flat["__cases__"] = np.array(sorted(cases))
np.savez_compressed(OUT, **flat)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
