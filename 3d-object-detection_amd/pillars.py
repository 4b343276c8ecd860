"""point pillars data prep functions

Drop-in for the reference's pybind11 module ``pillars`` (/root/reference
data/pillars.cpp:429-435): the same two functions with the same positional
signatures and in-place semantics, backed by the HIP kernels of libpp_hip.so
through its C ABI (include/pp_hip.h).  ``data/dataset.py:6`` imports it as
``data.pillars``; ``utils/box_utils.py:11`` likewise.

Deliberate, documented tightenings over the reference:
  * output arrays must be genuine writable float64 ndarrays (the reference
    silently writes into a forcecast temporary and the results vanish);
  * a wrongly wound box raises ValueError instead of ``std::exit(1)``
    (pillars.cpp:166-169);
  * pillars are emitted in a deterministic order (``ORDER``) instead of
    boost::unordered_map iteration order (pillars.cpp:335);
  * NaN coordinates are dropped (the reference feeds them to floor()).

HIP is initialised lazily, per process, on first call -- never at import -- so
the module survives ``DataLoader`` worker start-up (train.py:120-121); use the
``spawn`` start method, a HIP context does not survive ``fork``.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib

__all__ = ["create_pillars", "make_ious"]

#: pillar emission order used by create_pillars (0 row-major, 1 scrambled = default: the
#: stand-in for the reference's hash-map iteration order, pillars.cpp:335)
ORDER = int(os.environ.get("PP_PILLAR_ORDER", _lib.ORDER_SCRAMBLED))
#: HIP device used by this module's lazily created context
DEVICE = int(os.environ.get("PP_HIP_DEVICE", "0"))

_ctx = None
_ctx_pid = None
# The module's context (staging buffers, one stream) serves ONE call at a time; ctypes releases the GIL around a foreign
# call, so two Python threads could otherwise be inside it at once.  (The reference's module holds the GIL for the whole
# call, pillars.cpp:429-435: its callers never overlap either.)
_call_lock = threading.Lock()


def _context():
    global _ctx, _ctx_pid
    if _ctx is None or _ctx_pid != os.getpid():
        _ctx = _lib.Context(DEVICE)
        _ctx_pid = os.getpid()
    return _ctx


def _i64(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def _in_f64(a, name, ndim):
    a = np.asarray(a)
    if a.dtype != np.float64:  # pybind11 array_t<double> forcecast on an input: harmless
        a = a.astype(np.float64)
    if a.ndim != ndim:
        raise IndexError(f"{name}: index dimension mismatch; {ndim} expected, got {a.ndim}")
    return a


def _out_f64(a, name, ndim):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError(f"{name} must be a float64 numpy.ndarray: the reference would write "
                        "into a converted temporary and lose every result")
    if not a.flags.writeable:
        raise TypeError(f"{name} must be writable")
    if a.ndim != ndim:
        raise IndexError(f"{name}: index dimension mismatch; {ndim} expected, got {a.ndim}")
    return a


def create_pillars(points, tensor, indices, max_points_per_pillar, max_pillars,
                   x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max, canvas_height):
    """pillars

    create_pillars(points[n,>=4], tensor[P,N,9], indices[P,3], max_points_per_pillar,
    max_pillars, x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max,
    canvas_height) -> None   (data/pillars.cpp:236-249, exported :433)

    Fills ``tensor`` with the 9 features of the first ``max_points_per_pillar``
    points of up to ``max_pillars`` pillars and ``indices`` with
    ``[1, canvas_x, canvas_y]``; nothing else is touched (the caller supplies
    zeros, data/dataset.py:89-90).
    """
    points = _in_f64(points, "points", 2)
    tensor = _out_f64(tensor, "tensor", 3)
    indices = _out_f64(indices, "indices", 2)
    if points.shape[0] > 0 and points.shape[1] < 4:
        raise IndexError("points: index 3 is out of bounds for axis 1")
    prm = _lib.make_voxel_params(max_points_per_pillar, max_pillars, x_step, y_step, x_min, y_min,
                                 z_min, x_max, y_max, z_max, canvas_height, ORDER)
    ncell = ctypes.c_int64()
    with _call_lock:
        rc = _lib.lib().pp_create_pillars_f64(
            _context().handle, points.ctypes.data, points.shape[0], points.strides[0],
            points.strides[1], tensor.ctypes.data, _i64(tensor.shape), _i64(tensor.strides),
            indices.ctypes.data, _i64(indices.shape), _i64(indices.strides),
            ctypes.byref(prm), ctypes.byref(ncell))
        _lib.check(rc, "create_pillars")
    return None


def make_ious(a_corners, g_corners, a_centers, g_centers, ious):
    """ious

    make_ious(a_corners[A,4,2], g_corners[G,4,2], a_centers[A,>=2], g_centers[G,>=2],
    ious[A,G]) -> None   (data/pillars.cpp:400-404, exported :432)

    Every ``ious[i,j]`` is written: 0 when the centres are more than 10 cells
    apart in x or y, else the rotated-quad IoU (anchor corners counter-clockwise,
    ground-truth corners clockwise).
    """
    a_corners = _in_f64(a_corners, "a_corners", 3)
    g_corners = _in_f64(g_corners, "g_corners", 3)
    a_centers = _in_f64(a_centers, "a_centers", 2)
    g_centers = _in_f64(g_centers, "g_centers", 2)
    ious = _out_f64(ious, "ious", 2)
    A, G = a_corners.shape[0], g_corners.shape[0]
    if A == 0 or G == 0:
        return None
    if (a_corners.shape[1] < 4 or a_corners.shape[2] < 2 or g_corners.shape[1] < 4
            or g_corners.shape[2] < 2 or a_centers.shape[0] < A or a_centers.shape[1] < 2
            or g_centers.shape[0] < G or g_centers.shape[1] < 2
            or ious.shape[0] < A or ious.shape[1] < G):
        raise IndexError("make_ious: index out of bounds for the given array shapes")
    with _call_lock:
        rc = _lib.lib().pp_make_ious_f64(
            _context().handle, a_corners.ctypes.data, A, _i64(a_corners.strides),
            g_corners.ctypes.data, G, _i64(g_corners.strides),
            a_centers.ctypes.data, _i64(a_centers.strides),
            g_centers.ctypes.data, _i64(g_centers.strides),
            ious.ctypes.data, _i64(ious.strides))
        _lib.check(rc, "make_ious")
    return None
