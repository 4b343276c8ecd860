"""Reference-run vectors of the box decode (tests/golden/decode_ref_golden.npz, made by
tests/golden/make_decode_ref_golden.py by importing /root/reference/evaluate.py and CALLING its
``make_pred_boxes`` (:33-89) and ``move_box_to_car_space`` (:91-125) on field-holder stand-ins for the absent
lyft ``Box`` / pyquaternion ``Quaternion``).

What they pin: the decode arithmetic (anchor diagonal, offsets, exp, arcsin + anchor yaw, row flip, cell -> metre
scaling).  What they do not: the selection before it (sigmoid / tanh / class max / threshold / nms / first 100 are
inline in evaluate() and restated; torchvision's nms is absent) -- the kept anchors are the oracle's and enter as
inputs.  CPU: the oracle equals the reference's rows; GPU: the HIP decode equals them (kept ids exact, boxes 1e-5)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_ref_golden.npz")


def _cases():
    from pp_amd import boxes
    g = dict(np.load(GOLD))
    for name in sorted({k.split("/")[0] for k in g}):
        fm, canvas, step, x_min, thresh = g[f"{name}/geom"]
        acfg = boxes.AnchorConfig(int(fm), int(fm), 0.5, tuple(tuple(d) for d in g[f"{name}/dims"]),
                                  tuple(g[f"{name}/yaws_deg"]), tuple(g[f"{name}/zs"]))
        yield (name, acfg, boxes.make_anchors(acfg), g[f"{name}/cls"], g[f"{name}/reg"], g[f"{name}/kept"],
               g[f"{name}/boxes"], int(canvas), float(step), float(x_min), float(thresh))


def test_oracle_decode_equals_the_reference_run(oracle):
    n = 0
    for name, acfg, a, cls, reg, kept, want, canvas, step, x_min, thresh in _cases():
        b, k = oracle.postprocess(cls, reg, a["centers"], a["wlh"], a["yaw"], a["xy"], canvas, step, step, x_min, x_min,
                                  pos_thresh=thresh, nms_thresh=0.1)
        assert np.array_equal(k.astype(np.int32), kept), name            # (the selection: an oracle-drift pin)
        assert np.allclose(b, want, rtol=1e-6, atol=1e-6), name          # the reference's own decode
        assert np.array_equal(b[:, 8], want[:, 8])
        n += len(kept)
    assert n > 50


@pytest.mark.gpu
def test_hip_decode_equals_the_reference_run(gpu):
    import torch
    from pp_amd.postprocess import Detector
    for name, acfg, a, cls, reg, kept, want, canvas, step, x_min, thresh in _cases():
        det = Detector(a, acfg, canvas, step, step, x_min, x_min, pos_thresh=thresh, nms_thresh=0.1, device=gpu)
        boxes_d, kept_d, count_d = det(torch.from_numpy(cls).to(gpu), torch.from_numpy(reg).to(gpu))
        torch.cuda.synchronize()
        n = int(count_d.item())
        assert n == len(kept) and np.array_equal(kept_d.cpu().numpy()[:n], kept), name
        got = boxes_d.cpu().numpy()[:n]
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5), name
        assert np.array_equal(got[:, 8], want[:, 8])
