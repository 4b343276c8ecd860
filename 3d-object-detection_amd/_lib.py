"""ctypes binding of libpp_hip.so (the C ABI declared in include/pp_hip.h).

The library is the product: if it is missing this module raises -- there is no
Python or CPU fallback for any compute entry point.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("PP_HIP_LIB") or os.path.join(_HERE, "libpp_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f)
           for f in ("pp_runtime.hip", "pp_voxelize.hip", "pp_iou.hip", "pp_ingest.hip", "pp_decode.hip", "pp_epilogue.hip", "pp_pfn.hip", "pp_pfn_train.hip", "pp_bn_train.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "pp_common.h"), os.path.join(_ROOT, "include", "pp_hip.h")]

PP_OK, PP_ERR_INDEX, PP_ERR_VALUE, PP_ERR_WINDING = 0, -2, -3, -4
PP_ERR_NOMEM, PP_ERR_HIP, PP_ERR_INTERNAL = -5, -6, -7
ORDER_ROW_MAJOR, ORDER_SCRAMBLED = 0, 1
NUM_FEATURES = 9
MAX_BATCH = 32
MAX_INGEST_SWEEPS = 16       # PP_MAX_INGEST_SWEEPS
KERNEL_SPLIT, KERNEL_TILE, KERNEL_EMIT = 0, 1, 2

EXPORTS = [
    "pp_last_error", "pp_version", "pp_device_count", "pp_ctx_create", "pp_ctx_destroy",
    "pp_voxelize_reserve", "pp_voxelize_dev", "pp_voxelize_step_dev", "pp_voxelize_step_pfn_canvas_dev", "pp_voxelize_step_kernel_name", "pp_voxelize_step_reset", "pp_subtract_mean_dev", "pp_voxelize_pfn_dev", "pp_voxelize_pfn_canvas_dev", "pp_voxelize_pfn_canvas_reuse_dev", "pp_pfn_dense_dev", "pp_scatter_canvas_dev", "pp_pfn_train_stats_dev", "pp_pfn_train_backward_dev", "pp_create_pillars_f64", "pp_make_ious_f64",
    "pp_iou_check", "pp_make_ious_dev", "pp_assign_targets_dev", "pp_assign_targets_grid_dev", "pp_assign_targets_batch_dev", "pp_assign_targets_grid_batch_dev", "pp_ingest_dev", "pp_ingest_sweeps_dev", "pp_decode_dev", "pp_decode_strided_dev", "pp_decode_batch_dev", "pp_bias_relu_bn_dev", "pp_bias_relu_bn_nhwc_dev", "pp_relu_bn_train_fwd_dev", "pp_relu_bn_train_bwd_dev", "pp_ctx_set_timing",
    "pp_ctx_read_emit_ms", "pp_ctx_read_kernel_ms", "pp_voxelize_check", "pp_host_pool_selftest",
]


class VoxelParams(ctypes.Structure):
    """pp_voxel_params_t: the scalar arguments of create_pillars
    (/root/reference data/pillars.cpp:239-249)."""
    _fields_ = [
        ("max_points_per_pillar", ctypes.c_int32), ("max_pillars", ctypes.c_int32),
        ("x_step", ctypes.c_double), ("y_step", ctypes.c_double),
        ("x_min", ctypes.c_double), ("y_min", ctypes.c_double), ("z_min", ctypes.c_double),
        ("x_max", ctypes.c_double), ("y_max", ctypes.c_double), ("z_max", ctypes.c_double),
        ("canvas_height", ctypes.c_double),
        ("order", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class DecodeParams(ctypes.Structure):
    _fields_ = [("fm_height", ctypes.c_int32), ("fm_width", ctypes.c_int32),
                ("anchors_per_cell", ctypes.c_int32), ("num_classes", ctypes.c_int32),
                ("pos_thresh", ctypes.c_double), ("nms_thresh", ctypes.c_double),
                ("max_out", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("canvas_height", ctypes.c_double), ("x_step", ctypes.c_double),
                ("y_step", ctypes.c_double), ("x_min", ctypes.c_double), ("y_min", ctypes.c_double)]


class TargetParams(ctypes.Structure):
    _fields_ = [("pos_thresh", ctypes.c_double), ("canvas_height", ctypes.c_double),
                ("num_classes", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into libpp_hip.so (in-tree)."""
    newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off",            # every f64 product/sum rounds separately
           "--offload-arch=gfx950",
           "-I" + os.path.join(_ROOT, "include"),
           "-Wl,-rpath,/opt/rocm/lib",
           *SOURCES, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


VARIANTS = {
    # test infrastructure (tests/test_gpu_handoff.py): the target kernels' hand-off to a sample's last workgroup in the
    # architecturally guaranteed release / acquire form, to be compared bit for bit with the shipped write-through form
    "strict": ["-DPP_STRICT_HANDOFF"],
}


def variant_path(name):
    return os.path.join(_HERE, "variants", f"libpp_hip_{name}.so")


def build_variant(name, force=False, verbose=False):
    """Compile the whole library with the variant's defines into variants/libpp_hip_<name>.so (in-tree, git-ignored;
    travels to the GPU box like libpp_hip.so).  Never loaded by the product."""
    out = variant_path(name)
    newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "--offload-arch=gfx950",
           "-I" + os.path.join(_ROOT, "include"), "-Wl,-rpath,/opt/rocm/lib", *VARIANTS[name],
           *SOURCES, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def pybind_module_path():
    """Where build_pybind_module() puts the reference-style extension module ``pillars``."""
    import sysconfig
    return os.path.join(_HERE, "native", "pillars" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind_module(force=False, verbose=False):
    """Compile csrc/pillars_module.cpp (the reference's pybind11 module surface,
    data/pillars.cpp:429-435, on the C ABI) with g++ and link it against libpp_hip.so."""
    import pybind11
    import sysconfig
    src = os.path.join(_HERE, "csrc", "pillars_module.cpp")
    out = pybind_module_path()
    build()
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_ROOT, "include", "pp_hip.h")))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.join(_ROOT, "include"), src, "-o", out + ".tmp",
           "-L" + _HERE, "-lpp_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


_lib = None
_lock = threading.Lock()
_variants = {}


def lib():
    """Load libpp_hip.so once per process; never initialises HIP by itself."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        _lib = _load(LIB_PATH)
    return _lib


def variant_lib(name):
    """A build_variant() library as a second, independent instance in this process (its own code object, its own
    contexts): ``Context(device, lib=variant_lib("strict"))``.  Tests only."""
    with _lock:
        if name not in _variants:
            _variants[name] = _load(variant_path(name))
    return _variants[name]


def _load(path):
    """Load one build of the library and declare the C ABI's argument types on it."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the host drop-in
        pass
    L = ctypes.CDLL(path)
    c_int, i64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    pi64 = ctypes.POINTER(ctypes.c_int64)
    L.pp_last_error.restype = ctypes.c_char_p
    L.pp_version.restype = ctypes.c_char_p
    L.pp_device_count.restype = c_int
    L.pp_ctx_create.argtypes = [c_int, ctypes.POINTER(vp)]
    L.pp_ctx_destroy.argtypes = [vp]
    L.pp_ctx_destroy.restype = None
    L.pp_voxelize_reserve.argtypes = [vp, c_int, i64, ctypes.POINTER(VoxelParams)]
    L.pp_voxelize_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                  ctypes.POINTER(VoxelParams), vp, vp, vp]
    L.pp_voxelize_step_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                       ctypes.POINTER(VoxelParams), vp, vp, vp, ctypes.POINTER(c_int)]
    L.pp_voxelize_step_pfn_canvas_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                                  ctypes.POINTER(VoxelParams), vp, c_int, vp, c_int, c_int,
                                                  c_int, vp, vp, vp, vp, c_int, ctypes.POINTER(c_int)]
    L.pp_voxelize_step_kernel_name.argtypes = [ctypes.POINTER(VoxelParams), c_int, ctypes.c_char_p, c_int]
    L.pp_voxelize_step_reset.argtypes = [vp]
    L.pp_subtract_mean_dev.argtypes = [vp, vp, vp, c_int, i64, vp]
    L.pp_voxelize_pfn_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                      ctypes.POINTER(VoxelParams), vp, c_int, vp, vp, vp]
    L.pp_voxelize_pfn_canvas_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                             ctypes.POINTER(VoxelParams), vp, c_int, vp, c_int, c_int,
                                             c_int, vp, vp]
    L.pp_voxelize_pfn_canvas_reuse_dev.argtypes = [vp, vp, vp, i64, ctypes.POINTER(ctypes.c_int32), c_int,
                                                   ctypes.POINTER(VoxelParams), vp, c_int, vp, c_int, c_int,
                                                   c_int, vp, vp, vp]
    L.pp_pfn_dense_dev.argtypes = [vp, vp, vp, c_int, c_int, c_int, vp, c_int, vp]
    L.pp_scatter_canvas_dev.argtypes = [vp, vp, vp, vp, c_int, c_int, c_int, vp, c_int, c_int, c_int]
    L.pp_pfn_train_stats_dev.argtypes = [vp, vp, vp, c_int, c_int, c_int, vp, c_int, vp]
    L.pp_pfn_train_backward_dev.argtypes = [vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp]
    L.pp_create_pillars_f64.argtypes = [vp, vp, i64, i64, i64, vp, pi64, pi64, vp, pi64, pi64,
                                        ctypes.POINTER(VoxelParams), pi64]
    L.pp_make_ious_f64.argtypes = [vp, vp, i64, pi64, vp, i64, pi64, vp, pi64, vp, pi64, vp, pi64]
    L.pp_iou_check.argtypes = [vp, vp]
    L.pp_make_ious_dev.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, i64, i64, vp]
    L.pp_assign_targets_dev.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                        vp, ctypes.POINTER(TargetParams), vp, vp]
    L.pp_assign_targets_grid_dev.argtypes = [vp, vp, c_int, c_int, ctypes.c_double, c_int, vp, i64, vp, vp,
                                             vp, vp, vp, vp, ctypes.POINTER(TargetParams), vp, vp]
    pi32 = ctypes.POINTER(ctypes.c_int32)
    L.pp_assign_targets_batch_dev.argtypes = [vp, vp, ctypes.c_int32, pi32, i64, vp, vp, vp, vp, vp, vp, vp,
                                              vp, vp, vp, ctypes.POINTER(TargetParams), vp, vp]
    L.pp_assign_targets_grid_batch_dev.argtypes = [vp, vp, ctypes.c_int32, pi32, c_int, c_int, ctypes.c_double,
                                                   c_int, vp, vp, vp, vp, vp, vp, vp,
                                                   ctypes.POINTER(TargetParams), vp, vp]
    L.pp_ingest_dev.argtypes = [vp, vp, vp, i64, c_int, ctypes.POINTER(ctypes.c_double),
                                ctypes.c_double, vp]
    L.pp_ingest_sweeps_dev.argtypes = [vp, vp, c_int, vp, vp, c_int, vp, ctypes.c_double, vp]
    L.pp_decode_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(DecodeParams), vp, vp, vp]
    L.pp_decode_strided_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp,
                                        ctypes.POINTER(DecodeParams), vp, vp, vp]
    L.pp_decode_batch_dev.argtypes = [vp, vp, c_int, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp,
                                      ctypes.POINTER(DecodeParams), vp, vp, vp]
    L.pp_bias_relu_bn_dev.argtypes = [vp, vp, vp, i64, c_int, i64, vp, vp, i64, i64]
    L.pp_bias_relu_bn_nhwc_dev.argtypes = [vp, vp, vp, i64, c_int, vp, vp, i64, i64]
    dbl = ctypes.c_double
    L.pp_relu_bn_train_fwd_dev.argtypes = [vp, vp, vp, vp, i64, c_int, i64, vp, vp, dbl, dbl, vp, vp, vp, vp, vp]
    L.pp_relu_bn_train_bwd_dev.argtypes = [vp, vp, vp, vp, vp, i64, i64, c_int, i64, vp, vp, vp, vp, vp, vp, vp]
    L.pp_ctx_set_timing.argtypes = [vp, c_int]
    L.pp_ctx_read_emit_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), c_int,
                                      ctypes.POINTER(c_int)]
    L.pp_ctx_read_kernel_ms.argtypes = [vp, c_int, ctypes.POINTER(ctypes.c_float), c_int,
                                        ctypes.POINTER(c_int)]
    L.pp_voxelize_check.argtypes = [vp, vp]
    L.pp_host_pool_selftest.argtypes = [c_int, c_int, c_int]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("pp_last_error", "pp_version", "pp_ctx_destroy"):
            fn.restype = c_int
    return L


class PPError(RuntimeError):
    pass


def check(rc, what="", msg=None):
    """Map a PP_ERR_* code to the exception the reference's pybind11 module
    raises for the same condition (IndexError) or to a loud failure.  ``msg``: the error text, when the
    caller read pp_last_error() before making further library calls."""
    if rc == PP_OK:
        return
    msg = (msg if msg is not None else lib().pp_last_error()).decode("utf-8", "replace")
    msg = f"{what}: {msg}" if what else msg
    if rc == PP_ERR_INDEX:
        raise IndexError(msg)          # pybind11 index_error
    if rc == PP_ERR_VALUE:
        raise ValueError(msg)
    if rc == PP_ERR_WINDING:
        raise ValueError(msg)          # reference: std::exit(1), pillars.cpp:166-169
    if rc == PP_ERR_NOMEM:
        raise MemoryError(msg)
    raise PPError(f"{msg} (rc={rc})")


class Context:
    """Owner of one pp_ctx_t (scratch buffers for one device / one stream)."""

    def __init__(self, device=0, lib_=None):
        self._h = ctypes.c_void_p()
        self.lib = lib_ if lib_ is not None else lib()
        check(self.lib.pp_ctx_create(int(device), ctypes.byref(self._h)), "pp_ctx_create",
              None if lib_ is None else self.lib.pp_last_error())
        self.device = int(device)
        self._pid = os.getpid()

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) and self._h.value and self._pid == os.getpid():
            self.lib.pp_ctx_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_voxel_params(max_points_per_pillar, max_pillars, x_step, y_step, x_min, y_min, z_min,
                      x_max, y_max, z_max, canvas_height, order=ORDER_SCRAMBLED):
    return VoxelParams(int(max_points_per_pillar), int(max_pillars), float(x_step), float(y_step),
                       float(x_min), float(y_min), float(z_min), float(x_max), float(y_max),
                       float(z_max), float(canvas_height), int(order), 0)
