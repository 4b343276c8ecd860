"""Generates tests/golden/decode_ref_golden.npz by running the REFERENCE's own ``make_pred_boxes`` and
``move_box_to_car_space`` (/root/reference/evaluate.py:33-89, 91-125) in this container (container-only; the
reference never travels to the GPU box, the fixture does).

evaluate.py imports packages that are absent from the image.  None carries arithmetic on the two functions
run here, so the import is satisfied with arithmetic-free stand-ins: ``easydict`` (attribute dict);
``torchvision.ops.nms``, ``lyft_dataset_sdk.eval...get_average_precisions``, ``data.dataset.PPDataset`` bound to
a class that raises when touched (never touched here); ``pyquaternion.Quaternion`` and
``lyft_dataset_sdk...Box`` bound to FIELD HOLDERS: they store their constructor arguments (Quaternion: axis /
radians or the 4-list it was built from; Box: center, size -> wlh, orientation, name, score, token) and compute
nothing.  The anchor objects are the attribute holders of make_targets_ref_golden.py.

WHAT THIS PINS: the decode arithmetic of make_pred_boxes (anchor diagonal, centre offsets, exp of the size
offsets, arcsin + anchor yaw, class name lookup) and of move_box_to_car_space (row flip, cell -> metre scaling of
centre and size), executed by the reference's own source.  WHAT IT DOES NOT PIN: the lines of evaluate() that
precede them (sigmoid, tanh, class max, threshold, box_nms, first 100: inline in evaluate(), restated in
oracle.postprocess), torchvision's nms (absent), and the yaw -> quaternion -> yaw round trip of the SDK (the
holder keeps the yaw).  The selection (which anchors are kept) is therefore an INPUT here: the oracle's.

Run:  python tests/golden/make_decode_ref_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "decode_ref_golden.npz")

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
from pp_amd import boxes  # noqa: E402


class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError("absent third-party class used: this generator must not need it")


class QuaternionFields:
    """stores what it is built from; computes nothing"""
    def __init__(self, *args, axis=None, radians=None, degrees=None):
        if args:                                   # Quaternion(list(q)) in move_box_to_car_space
            (fields,) = args
            self.fields = tuple(fields)
        else:
            self.fields = ("axis-angle", tuple(axis), radians, degrees)

    def __iter__(self):
        return iter(self.fields)

    @property
    def yaw(self):
        assert self.fields[0] == "axis-angle" and self.fields[1] == (0, 0, 1) and self.fields[3] is None
        return self.fields[2]


class BoxFields:
    def __init__(self, center=None, size=None, orientation=None, name=None, score=None, token=None):
        self.center, self.wlh, self.orientation = np.array(center), np.array(size), orientation
        self.name, self.score, self.token = name, score, token


def _stub(name, **names):
    m = types.ModuleType(name)
    m.__dict__.update(names)
    sys.modules[name] = m
    return m


_stub("pyquaternion", Quaternion=QuaternionFields)
_stub("lyft_dataset_sdk")
_stub("lyft_dataset_sdk.utils")
_stub("lyft_dataset_sdk.utils.data_classes", Box=BoxFields, LidarPointCloud=_Absent)
_stub("lyft_dataset_sdk.eval")
_stub("lyft_dataset_sdk.eval.detection")
_stub("lyft_dataset_sdk.eval.detection.mAP_evaluation", get_average_precisions=_Absent)
_stub("torchvision")
_stub("torchvision.ops", nms=_Absent)
_ds = types.ModuleType("data.dataset")
_ds.PPDataset = _Absent
sys.modules["data.dataset"] = _ds
sys.path.insert(0, REF)
from config import cfg  # noqa: E402
import data as _ref_data  # noqa: E402
_ref_data.dataset = _ds
import evaluate as ref  # noqa: E402   the reference's own source, unmodified

assert ref.make_pred_boxes.__code__.co_filename.startswith(REF)


class _Orientation:
    def __init__(self, yaw):
        self.yaw_pitch_roll = (float(yaw), 0.0, 0.0)


class _AnchorFields:
    def __init__(self, center, wlh, yaw):
        self.center, self.wlh, self.orientation = np.array(center, np.float64), np.array(wlh, np.float64), _Orientation(yaw)


out = {}
NAME_TO_IND = {v: int(k) for k, v in cfg.DATA.IND_TO_NAME.items()}
for case, (fm, canvas, step, x_min, seed, thresh) in {"map16": (16, 32, 0.2, -3.2, 77, 0.3),
                                                      "map24_default_set": (24, 48, 0.2, -4.8, 5, 0.3)}.items():
    if case == "map16":
        acfg = boxes.AnchorConfig(fm, fm)
    else:
        d = boxes.AnchorConfig.reference_default()
        acfg = boxes.AnchorConfig(fm, fm, 0.5, d.dims, d.yaws_deg, d.zs)
    anch = boxes.make_anchors(acfg)
    rng = np.random.default_rng(seed)
    cls = rng.normal(-1.6, 1.5, (acfg.per_cell * 9, fm, fm)).astype(np.float32)
    reg = rng.normal(0.0, 0.3, (acfg.per_cell * 8, fm, fm)).astype(np.float32)
    want, kept = O.postprocess(cls, reg, anch["centers"], anch["wlh"], anch["yaw"], anch["xy"], canvas, step, step,
                               x_min, x_min, pos_thresh=thresh, nms_thresh=0.1)
    print(case, 'kept', len(kept))
    assert len(kept) >= 5
    # the inline lines of evaluate() before the two functions (evaluate.py:231-235), restated -- NOT reference-run
    cls_t = torch.sigmoid(torch.from_numpy(cls).permute(1, 2, 0).reshape(-1, 9))
    reg_t = torch.from_numpy(reg).permute(1, 2, 0).reshape(-1, 8).clone()
    reg_t[..., 6] = torch.tanh(reg_t[..., 6])
    scores, classes = torch.max(cls_t, dim=-1)
    cfg.DATA.CANVAS_HEIGHT, cfg.DATA.X_STEP, cfg.DATA.Y_STEP = canvas, step, step
    cfg.DATA.X_MIN, cfg.DATA.Y_MIN = x_min, x_min
    a_list = [_AnchorFields(c, w, y) for c, w, y in zip(anch["centers"], anch["wlh"], anch["yaw"])]
    pred = ref.make_pred_boxes(torch.from_numpy(kept), a_list, reg_t, classes, scores, "tok")   # reference-run
    rows = np.zeros((len(pred), 9))
    for r, b in enumerate(pred):
        car = ref.move_box_to_car_space(b)                                                     # reference-run
        yaw = QuaternionFields(list(car.orientation)).fields[2]
        assert car.token == "tok" and yaw == b.orientation.yaw
        rows[r] = [*car.center, *car.wlh, yaw, car.score, NAME_TO_IND[car.name]]
    assert np.allclose(rows, want, rtol=1e-6, atol=1e-6), np.abs(rows - want).max()
    out.update({f"{case}/cls": cls, f"{case}/reg": reg, f"{case}/kept": kept.astype(np.int32), f"{case}/boxes": rows,
                f"{case}/geom": np.array([fm, canvas, step, x_min, thresh]),
                f"{case}/dims": np.asarray(acfg.dims), f"{case}/yaws_deg": np.asarray(acfg.yaws_deg),
                f"{case}/zs": np.asarray(acfg.zs)})
    print(case, "kept", len(kept), "max |ref - oracle| =", float(np.abs(rows - want).max()))
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
