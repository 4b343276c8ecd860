"""GPU tests of the host drop-in module (the reference's pybind11 surface,
data/pillars.cpp:429-435) through pp_create_pillars_f64 / pp_make_ious_f64."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import grid_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["ctypes", "pybind11"])
def pillars(request):
    """Both host bindings of the two-function module surface: the ctypes mirror
    (pp_amd.pillars) and the pybind11 extension module `pillars` built from
    csrc/pillars_module.cpp -- the reference's own kind of module (pillars.cpp:429-435)."""
    if request.param == "ctypes":
        from pp_amd import pillars as mod
        return mod
    import importlib.util
    import pp_amd
    path = pp_amd._lib.build_pybind_module()
    spec = importlib.util.spec_from_file_location("pillars", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.__doc__ == "point pillars data prep functions"
    return mod


def _v1_points():
    # cell (1,2) x5 points, cell (0,0) x2 points, 3 boundary rejects
    return np.array([[1.1, 2.1, 0.0, 10], [0.5, 0.5, 0, 1], [1.2, 2.2, 0.1, 11], [1.3, 2.3, 0.2, 12],
                     [4.0, 1, 0, 0], [1, 1, 1.0, 0], [-0.001, 1, 0, 0], [1.5, 2.5, 0.3, 13],
                     [0.25, 0.75, 0.5, 2], [1.8, 2.8, 0.4, 14]])


def test_v1_hand_case_bit_exact(gpu, oracle, pillars):
    pts = _v1_points()
    T, I = np.zeros((4, 3, 9)), np.zeros((4, 3))
    assert pillars.create_pillars(pts, T, I, 3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4) is None
    Tr, Ir = np.zeros((4, 3, 9)), np.zeros((4, 3))
    oracle.create_pillars(pts, Tr, Ir, 3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    assert np.array_equal(T, Tr) and np.array_equal(I, Ir)
    # the row recorded by the survey probe
    assert np.allclose(T[0, 0], [1.1, 2.1, 0, 10, -0.1, -1.1, 0.28, 0.28, 0.2], atol=1e-12)
    assert list(I[0]) == [1, 1, 1] and list(I[1]) == [1, 0, 3]


def test_f64_points_strided_prefilled(gpu, oracle, pillars):
    """f64 (not f32-representable) coordinates, F-order strided view like
    data/dataset.py:88, outputs pre-filled with a sentinel stay untouched where
    the reference would not write, Python ints / numpy ints as scalars."""
    rng = np.random.default_rng(3)
    agg = rng.uniform(-12, 12, (4, 30000))          # [4, n] like agg_pc
    agg[2] = rng.uniform(-3, 3, 30000)
    pts = agg.transpose([1, 0])                     # strided view, dataset.py:88
    P, N = 20000, 12
    T = np.full((P, N, 9), 7.5)
    I = np.full((P, 3), -3.0)
    pillars.create_pillars(pts, T, I, N, P, .2, .2, -10, -10, -3, 10, 10, 3, np.int32(100))
    Tr = np.full((P, N, 9), 7.5)
    Ir = np.full((P, 3), -3.0)
    m = oracle.create_pillars(pts, Tr, Ir, N, P, .2, .2, -10, -10, -3, 10, 10, 3, 100)
    assert m < P
    assert np.array_equal(T, Tr) and np.array_equal(I, Ir)   # bit-exact f64 incl. sentinels
    assert (T == 7.5).any() and (I[m:] == -3.0).all()


def test_non_contiguous_outputs_and_f32_input(gpu, oracle, pillars):
    from pp_amd import synth
    pts32 = synth.lidar_like(5000, 8.0, 2)          # f32 input is force-cast (harmless)
    P, N = 4000, 8
    big = np.zeros((P, N, 18))
    T = big[:, :, ::2]                              # strided output view
    Ibig = np.zeros((P, 6))
    I = Ibig[:, ::2]
    pillars.create_pillars(pts32, T, I, N, P, .2, .2, -8, -8, -10, 8, 8, 10, 80)
    Tr, Ir = np.zeros((P, N, 9)), np.zeros((P, 3))
    oracle.create_pillars(pts32.astype(np.float64), Tr, Ir, N, P, .2, .2, -8, -8, -10, 8, 8, 10, 80)
    assert np.array_equal(T, Tr) and np.array_equal(I, Ir)
    assert not big[:, :, 1::2].any() and not Ibig[:, 1::2].any()


def test_errors_match_reference_surface(gpu, oracle, pillars):
    pts = _v1_points()
    # non-f64 outputs are rejected loudly (the reference silently loses the writes)
    with pytest.raises(TypeError):
        pillars.create_pillars(pts, np.zeros((4, 3, 9), np.float32), np.zeros((4, 3)),
                               3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    with pytest.raises(TypeError):
        pillars.create_pillars(pts, np.zeros((4, 3, 9)), np.zeros((4, 3), np.int64),
                               3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    # undersized tensor -> IndexError after the in-range part was written (pybind11 .mutable_at)
    T, I = np.zeros((1, 3, 9)), np.zeros((4, 3))
    with pytest.raises(IndexError):
        pillars.create_pillars(pts, T, I, 3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    Tr, Ir = np.zeros((1, 3, 9)), np.zeros((4, 3))
    with pytest.raises(IndexError):
        oracle.create_pillars(pts, Tr, Ir, 3, 4, 1, 1, 0, 0, -1, 4, 4, 1, 4)
    assert np.array_equal(T, Tr) and np.array_equal(I, Ir) and T.any()
    with pytest.raises(ValueError):
        pillars.create_pillars(pts, np.zeros((4, 3, 9)), np.zeros((4, 3)), 3, 4, 0.0, 1, 0, 0, -1, 4, 4, 1, 4)


def test_max_pillars_and_max_points_caps(gpu, oracle, pillars):
    from pp_amd import synth
    pts = synth.lidar_like(20000, 10.0, 9).astype(np.float64)
    for P, N in ((50, 3), (0, 5), (5, 0), (100000, 1000)):
        shape_p = max(P, 1) if P < 100000 else 12000
        T, I = np.zeros((shape_p, max(N, 1), 9)), np.zeros((shape_p, 3))
        Tr, Ir = T.copy(), I.copy()
        pillars.create_pillars(pts, T, I, N, min(P, shape_p), *grid_args(10.0, 0.2))
        oracle.create_pillars(pts, Tr, Ir, N, min(P, shape_p), *grid_args(10.0, 0.2))
        assert np.array_equal(T, Tr) and np.array_equal(I, Ir), (P, N)


def test_make_ious_host_dropin(gpu, oracle, pillars):
    from pp_amd import boxes, synth
    anchors = boxes.make_anchors(boxes.AnchorConfig(60, 60))
    gt = synth.gt_boxes(12, 120, seed=4, margin=15.0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 120)
    A, G = len(anchors["corners"]), 12
    ious = np.full((A, G), -5.0)
    assert pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious) is None
    ref = np.full((A, G), -5.0)
    oracle.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ref)
    assert np.array_equal(ious, ref)            # bit-exact: same f64 operation sequence
    assert (ious > 0.6).any() and (ious == 0).sum() > 0.9 * A * G
    # strided / transposed output and f32 output
    iousT = np.zeros((G, A)).T
    pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, iousT)
    assert np.array_equal(iousT, ref)
    with pytest.raises(TypeError):
        pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img,
                          np.zeros((A, G), np.float32))
    # wrong winding raises instead of std::exit(1) (pillars.cpp:166-169)
    with pytest.raises(ValueError):
        pillars.make_ious(anchors["corners"], k_img[:, ::-1].copy(), anchors["centers"], c_img,
                          np.zeros((A, G)))


def test_make_ious_anchors_stay_resident_and_follow_in_place_edits(gpu, oracle, pillars):
    """pp_make_ious_f64 keeps the previous call's anchors on the device and uploads them again only when the
    gather finds a changed bit.  The reference reads its arguments afresh on every call (pillars.cpp:385-420), so
    every kind of change between two calls must show in the result: one corner edited in place, one centre edited in
    place, another anchor count, strided (non-dense) anchor arrays, a -0.0 for a 0.0, and the same set again."""
    from pp_amd import boxes, synth
    anchors = boxes.make_anchors(boxes.AnchorConfig(40, 40))
    ac, an = anchors["corners"].copy(), anchors["centers"].copy()
    gt = synth.gt_boxes(9, 80, seed=11, margin=10.0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 80)
    A, G = len(ac), 9

    def both(ac_, an_, k_=k_img, c_=c_img):
        out, ref = np.full((len(ac_), len(k_)), -2.0), np.full((len(ac_), len(k_)), -2.0)
        pillars.make_ious(ac_, k_, an_, c_, out)
        oracle.make_ious(ac_, k_, an_, c_, ref)
        assert np.array_equal(out, ref)
        return out

    first = both(ac, an)
    assert np.array_equal(both(ac, an), first)                 # resident: nothing uploaded, same bits
    hit = int(np.argmax(first.max(axis=1)))                    # an anchor that overlaps a box
    ac[hit] += 0.25                                            # one anchor's corners, in place
    moved = both(ac, an)
    assert not np.array_equal(moved[hit], first[hit])
    an[hit] += 500.0                                           # its centre far away: the distance gate drops it
    gated = both(ac, an)
    assert not gated[hit].any() and moved[hit].any()
    an[hit] -= 500.0
    ac[hit] -= 0.25
    assert np.array_equal(both(ac, an), first)                 # and back
    # fewer anchors, then the full set again; then strided views of wider arrays (the element-wise gather)
    both(ac[: A // 3], an[: A // 3])
    assert np.array_equal(both(ac, an), first)
    wide_c, wide_n = np.zeros((A, 4, 4)), np.zeros((A, 6))
    wide_c[:, :, ::2], wide_n[:, ::3] = ac, an[:, :2]
    assert np.array_equal(both(wide_c[:, :, ::2], wide_n[:, ::3]), first)
    wide_c[hit, :, ::2] += 0.125
    assert not np.array_equal(both(wide_c[:, :, ::2], wide_n[:, ::3])[hit], first[hit])
    # a sign bit alone is a change (compared as bits, not as numbers)
    z = ac.copy()
    z[0, 0, 0] = 0.0
    a0 = both(z, an)
    z[0, 0, 0] = -0.0
    assert np.array_equal(both(z, an), a0)
    # other ground truths against resident anchors
    gt2 = synth.gt_boxes(5, 80, seed=12, margin=10.0)
    c2, k2 = boxes.boxes_to_image_space(gt2["centers"], gt2["wlh"], gt2["yaw"], 80)
    both(z, an, k2, c2)
    both(ac, an, k2, c2)


_DATA_PILLARS_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])              # the reference checkout's root: `data` is a package
import data.pillars as pillars               # data/dataset.py:6
assert pillars.__doc__ == "point pillars data prep functions" and pillars.__file__.startswith(sys.argv[1])
assert "pp_amd" not in sys.modules and "torch" not in sys.modules     # nothing but the two .so files
pts = np.array([[1.1, 2.1, 0.0, 10], [0.5, 0.5, 0, 1], [1.2, 2.2, 0.1, 11], [1.3, 2.3, 0.2, 12],
                [4.0, 1, 0, 0], [1, 1, 1.0, 0], [-0.001, 1, 0, 0], [1.5, 2.5, 0.3, 13],
                [0.25, 0.75, 0.5, 2], [1.8, 2.8, 0.4, 14]])
lidar_points = np.asfortranarray(pts)        # an F-order view like dataset.py:88
T, I = np.zeros((4, 3, 9)), np.zeros((4, 3)) # dataset.py:89-90
assert pillars.create_pillars(lidar_points, T, I, 3, 4, 1, 1, 0, 0, -1, 4, 4, 1, np.int32(4)) is None
assert np.allclose(T[0, 0], [1.1, 2.1, 0, 10, -0.1, -1.1, 0.28, 0.28, 0.2], atol=1e-12)   # SURVEY 8c V1
assert np.allclose(T[0, :, 0], [1.1, 1.2, 1.3]) and T[1, 0, 5] == 2.5 and not T[2:].any()
assert I.tolist() == [[1, 1, 1], [1, 0, 3], [0, 0, 0], [0, 0, 0]]
a = np.array([[[7., 3], [7, 7], [3, 7], [3, 3]]])           # 4x4 anchor, counter-clockwise
g = a[:, ::-1].copy()                                       # the same quad clockwise
ious = np.full((1, 1), -1.0)
pillars.make_ious(a, g, np.array([[5., 5, 0]]), np.array([[5., 5, 0]]), ious)             # box_utils.py:182
assert ious[0, 0] == 1.0
print("data.pillars ok")
"""


def test_import_as_data_pillars_in_the_reference_layout(gpu, tmp_path):
    """INTEGRATION.md section 2 / install_mods.sh:8-10: the built module is MOVED into the
    reference checkout's data/ package, next to libpp_hip.so, and the reference imports it as
    ``data.pillars`` (data/dataset.py:6, utils/box_utils.py:11).  A fresh child process with
    nothing of this repo on its path must be able to do exactly that: the $ORIGIN rpath of the
    extension module finds libpp_hip.so beside it."""
    import glob
    import os
    import shutil
    import subprocess
    import sys
    import pp_amd
    mod = pp_amd._lib.build_pybind_module()
    data = tmp_path / "data"
    data.mkdir()
    (data / "__init__.py").write_text("")
    shutil.copy(mod, data / os.path.basename(mod))
    shutil.copy(pp_amd._lib.LIB_PATH, data / "libpp_hip.so")
    assert len(glob.glob(str(data / "pillars*.so"))) == 1
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PP_HIP_LIB", "LD_LIBRARY_PATH")}
    r = subprocess.run([sys.executable, "-c", _DATA_PILLARS_CHILD, str(tmp_path)], cwd=str(tmp_path),
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "data.pillars ok" in r.stdout, r.stderr[-2000:]


_WORKER = r"""
import sys, time
import numpy as np
root, repo, wid, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, root)                       # the reference checkout's layout: data/pillars*.so
import data.pillars as pillars                 # data/dataset.py:6 -- lazily creates THIS process's context
assert pillars.__file__.startswith(root)
sys.path.insert(1, repo)                       # the checker (CPU oracle) comes from the repo
import pp_amd                                  # noqa: F401  (registers the package alias the oracle imports)
from pp_amd import boxes, synth
from oracle import oracle as O
half, step, P, N = 30.0, 0.25, 9000, 40
H = int(2 * half / step)
anchors = boxes.make_anchors(boxes.AnchorConfig(30, 30))
for it in range(iters):
    seed = 1000 * wid + it
    pts = synth.lidar_like(20000 + 500 * wid, half, seed).astype(np.float64)
    T, I = np.zeros((P, N, 9)), np.zeros((P, 3))                                  # dataset.py:89-90
    pillars.create_pillars(np.asfortranarray(pts), T, I, N, P, step, step, -half, -half, -10, half, half, 10, H)
    rT, rI = np.zeros((P, N, 9)), np.zeros((P, 3))
    O.create_pillars(pts, rT, rI, N, P, step, step, -half, -half, -10.0, half, half, 10.0, H, order=1)
    assert np.array_equal(I, rI) and np.array_equal(T, rT), (wid, it)
    gt = synth.gt_boxes(8, 60, seed, margin=8.0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 60)
    ious, ref = np.zeros((len(anchors["corners"]), 8)), np.zeros((len(anchors["corners"]), 8))
    pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious)  # box_utils.py:182
    O.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ref)
    assert np.array_equal(ious, ref), (wid, it)
print("worker", wid, "ok", iters)
"""


def test_concurrent_worker_processes_share_one_device(gpu, tmp_path):
    """The reference calls the module from DataLoader(num_workers=4) worker processes (train.py:120-121,
    data/dataset.py:92-97).  Here: three FRESH processes (spawned, never forked from a GPU-initialised one),
    each importing ``data.pillars`` from the reference-layout scratch directory, creating its own lazy
    context and looping create_pillars + make_ious on the SAME device at the same time; every result is
    checked against the CPU oracle inside the worker.  A worker started without a visible device fails
    loudly (no CPU fallback)."""
    import os
    import shutil
    import subprocess
    import sys
    import pp_amd
    from oracle import oracle as O
    assert hasattr(O, "create_pillars")
    mod = pp_amd._lib.build_pybind_module()
    data = tmp_path / "data"
    data.mkdir()
    (data / "__init__.py").write_text("")
    shutil.copy(mod, data / os.path.basename(mod))
    shutil.copy(pp_amd._lib.LIB_PATH, data / "libpp_hip.so")
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PP_HIP_LIB", "LD_LIBRARY_PATH")}
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, str(tmp_path), ROOT, str(w), "12"], cwd=str(tmp_path),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(3)]
    outs = [p.communicate(timeout=600) for p in procs]
    for w, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"worker {w} ok 12" in so, se[-2000:]
    # no device visible: the first call raises, naming the missing device and the absent fallback
    env_nodev = dict(env, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", _WORKER, str(tmp_path), ROOT, "9", "1"], cwd=str(tmp_path),
                       env=env_nodev, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no HIP device available" in r.stderr and "no CPU fallback" in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("canvas_height", [80, 200, 33])
def test_rectangular_offset_grid_host_module(gpu, oracle, pillars, canvas_height, monkeypatch):
    """The drop-in signature's nine independent grid scalars (pillars.cpp:236-249) through both host bindings:
    x_step = 0.25, y_step = 0.4, x in [-30, 50), y in [-12, 20), canvas_height equal to / larger than / smaller than the
    row count; f64 points that are NOT f32-representable, the strided view of data/dataset.py:88; overflow and no
    overflow: tensor and indices bit for bit the oracle's (the module's default order: scrambled)."""
    rng = np.random.default_rng(canvas_height)
    n = 40000
    agg = np.empty((4, n))
    agg[0], agg[1] = rng.normal(8.0, 22.0, n), rng.normal(3.0, 9.0, n)
    agg[2], agg[3] = rng.uniform(-3, 3, n), rng.uniform(0, 255, n)
    agg[0, :4000], agg[1, :4000] = rng.uniform(11.0, 12.5, 4000), rng.uniform(-2.0, -0.4, 4000)
    pts = agg.transpose([1, 0])
    g = (0.25, 0.4, -30.0, -12.0, -2.5, 50.0, 20.0, 2.5, canvas_height)
    for P, N in ((20000, 24), (2500, 24)):
        T, I = np.full((P, N, 9), 0.5), np.full((P, 3), -1.0)
        assert pillars.create_pillars(pts, T, I, N, P, *g) is None
        Tr, Ir = np.full((P, N, 9), 0.5), np.full((P, 3), -1.0)
        m = oracle.create_pillars(pts, Tr, Ir, N, P, *g, order=oracle.ORDER_SCRAMBLED)
        assert (m > P) == (P == 2500)
        assert np.array_equal(I, Ir) and np.array_equal(T, Tr)
        if canvas_height == 33:
            assert (I[:min(m, P), 2] < 0).any()


def test_f32_representable_points_take_the_f32_kernels_same_bits(gpu, oracle, pillars):
    """The host entry point sends a cloud whose every value survives double -> float -> double through the f32-input
    kernels (half the bytes over PCIe; what data/dataset.py:51-88 hands over IS f32 values widened by np.hstack) and
    any other cloud through the f64-input kernels.  Both must give the oracle's bits on the SAME f64 input -- also
    when exactly one value of 40 000 points is not an f32 value (the whole call then takes the f64 kernels), and with
    a NaN row and a -0.0 in the cloud (NaN points are dropped, the library's documented deviation)."""
    from pp_amd import synth
    pts = synth.lidar_like(40000, 20.0, 9).astype(np.float64)           # f32 values, widened: dataset.py:82
    pts[7] = [-0.0, 0.0, -0.0, 3.0]
    P, N = 9000, 20
    g = (0.2, 0.2, -20.0, -20.0, -10.0, 20.0, 20.0, 10.0, 200)
    variants = {"f32 values": pts.copy()}
    one = pts.copy()
    one[12345, 1] += 1e-12                                               # one value between two f32 values
    variants["one f64 value"] = one
    nan = pts.copy()
    nan[100, 0] = np.nan
    variants["a NaN"] = nan
    outs = {}
    for name, p in variants.items():
        agg = np.ascontiguousarray(p.T)
        T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
        pillars.create_pillars(agg.transpose([1, 0]), T, I, N, P, *g)
        Tr, Ir = np.zeros((P, N, 9)), np.zeros((P, 3))
        q = p if name != "a NaN" else np.delete(p, 100, axis=0)        # the oracle's NaN row would poison a cell
        oracle.create_pillars(q, Tr, Ir, N, P, *g, order=oracle.ORDER_SCRAMBLED)
        assert np.array_equal(I, Ir), name
        assert np.array_equal(T.view(np.uint64), Tr.view(np.uint64)), name     # bits, signed zeros included
        outs[name] = (T, I)
    # the two kernel families agree wherever the perturbed point does not live
    same = np.all(outs["f32 values"][1] == outs["one f64 value"][1])
    assert same and (outs["f32 values"][0] != outs["one f64 value"][0]).sum() <= 9 * N


def test_python_threads_calling_at_once_are_serialised(gpu, oracle, pillars):
    """Both bindings release the GIL around the C call (ctypes always does; the pybind11 module does so that its pool
    of host threads can run), and the module's one context serves one call at a time: calls from several Python threads
    must queue up, each getting its own cloud's result.  (The reference holds the GIL for the whole call,
    pillars.cpp:429-435: its callers never overlap.)"""
    import threading
    from pp_amd import synth
    P, N = 6000, 16
    g = (0.25, 0.25, -16.0, -16.0, -5.0, 16.0, 16.0, 5.0, 128)
    clouds = [synth.lidar_like(25000, 16.0, 50 + k).astype(np.float64) for k in range(4)]
    outs = [(np.zeros((P, N, 9)), np.zeros((P, 3))) for _ in clouds]
    errs = []

    def work(k):
        try:
            for _ in range(6):
                outs[k][0][:] = 0
                outs[k][1][:] = 0
                pillars.create_pillars(clouds[k], outs[k][0], outs[k][1], N, P, *g)
        except Exception as e:          # pragma: no cover
            errs.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k, c in enumerate(clouds):
        Tr, Ir = np.zeros((P, N, 9)), np.zeros((P, 3))
        oracle.create_pillars(c, Tr, Ir, N, P, *g, order=oracle.ORDER_SCRAMBLED)
        assert np.array_equal(outs[k][1], Ir) and np.array_equal(outs[k][0], Tr), k


_TRANSPORT_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import pp_amd
from pp_amd import pillars, synth
from oracle import oracle
P, N = 9000, 20
g = (0.2, 0.2, -20.0, -20.0, -10.0, 20.0, 20.0, 10.0, 200)
f32_valued = synth.lidar_like(40000, 20.0, 9).astype(np.float64)
f64_valued = f32_valued + 1e-9
far = f32_valued[:6000].copy()
far[:, 0] += 1000.0                                         # 6 000 points, every one outside the grid ...
few = far.copy()
few[[5, 4000, 5999]] = f32_valued[[5, 4000, 5999]]          # ... but three (two copy pieces, one of them empty)
for name, p in (("f32", f32_valued), ("f64", f64_valued), ("tiny", f32_valued[:37]), ("N0", f32_valued),
                ("none in range", far), ("three in range", few), ("three in range, f64", few + 1e-9)):
    n_ = 0 if name == "N0" else N
    for rep in range(3):                                    # the staging buffers are re-used call after call
        T, I = np.full((P, max(n_, 1), 9), 0.25), np.full((P, 3), -4.0)
        pillars.create_pillars(np.ascontiguousarray(p.T).transpose([1, 0]), T, I, n_, P, *g)
        Tr, Ir = np.full((P, max(n_, 1), 9), 0.25), np.full((P, 3), -4.0)
        oracle.create_pillars(p, Tr, Ir, n_, P, *g, order=oracle.ORDER_SCRAMBLED)
        assert np.array_equal(I, Ir), (name, rep)
        assert np.array_equal(T.view(np.uint64), Tr.view(np.uint64)), (name, rep)
    # a strided tensor: the element-wise scatter reads the same records
    big = np.zeros((P, N, 18))
    pillars.create_pillars(p, big[:, :, ::2], I, N, P, *g)
    Tr2 = np.zeros((P, N, 9))
    oracle.create_pillars(p, Tr2, Ir, N, P, *g, order=oracle.ORDER_SCRAMBLED)
    assert np.array_equal(big[:, :, ::2], Tr2) and not big[:, :, 1::2].any(), name
# make_ious: the records straight into pinned memory (any PP_DROPIN_DIRECT bit) or by copies (0)
from pp_amd import boxes
anchors = boxes.make_anchors(boxes.AnchorConfig(50, 50))
for seed, G in ((3, 11), (4, 1), (5, 30)):
    gt = synth.gt_boxes(G, 100, seed=seed, margin=12.0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 100)
    out, ref = np.full((len(anchors["corners"]), G), -1.0), np.full((len(anchors["corners"]), G), -1.0)
    pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, out)
    oracle.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ref)
    assert np.array_equal(out, ref) and (out > 0).any(), seed
try:
    pillars.make_ious(anchors["corners"], k_img[:, ::-1].copy(), anchors["centers"], c_img, out)
    raise SystemExit("wrong winding accepted")
except ValueError:
    pass
pillars.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, out)       # the error word was cleared
assert np.array_equal(out, ref)
print("transport ok")
"""


@pytest.mark.parametrize("direct,pack", [(0, 0), (0, 1), (3, 1), (4, 0), (7, 1)])
def test_host_transport_variants_give_the_same_bits(gpu, direct, pack, tmp_path):
    """pp_create_pillars_f64's ways over the link (read once per process from the environment, hence a child each):
    PP_DROPIN_DIRECT bit 1 -- k_split reads the gathered points from mapped pinned memory, no host-to-device copy; bit 2
    -- k_emit leaves descriptors and totals in mapped pinned memory as well; bit 4 -- the feature records come back
    through k_records_to_host, sized on the device; PP_DROPIN_PACK -- an f32-valued cloud's x, y, z, intensity travel as
    floats (56-byte records).  Every combination, for f32-valued and other clouds, a 37-point cloud, max_points 0 and
    a strided tensor: the oracle's bits (pillars.cpp:335-396).  pp_make_ious_f64 with its records appended straight into
    pinned memory (any bit) or copied (0): the oracle's matrix, a wrong winding raised and cleared (pillars.cpp:166-169)."""
    env = dict(os.environ, PP_DROPIN_DIRECT=str(direct), PP_DROPIN_PACK=str(pack))
    r = subprocess.run([sys.executable, "-c", _TRANSPORT_CHILD, ROOT], cwd=str(tmp_path), env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "transport ok" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])
