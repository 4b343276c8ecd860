"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/pp_hip.h declares; without a device the compute entry points fail
loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    import pp_amd
    L = pp_amd._lib.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"libpp_hip.so does not export {n}"
    assert sorted(pp_amd._lib.EXPORTS) == names
    assert b"gfx950" in L.pp_version()


def test_struct_layout_matches_header():
    import pp_amd
    assert ctypes.sizeof(pp_amd._lib.VoxelParams) == 4 + 4 + 9 * 8 + 4 + 4
    assert ctypes.sizeof(pp_amd._lib.TargetParams) == 8 + 8 + 4 + 4
    assert pp_amd._lib.VoxelParams.x_step.offset == 8 and pp_amd._lib.VoxelParams.order.offset == 80


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    import pp_amd
    from pp_amd import pillars
    import numpy as np
    assert pp_amd._lib.lib().pp_device_count() == 0
    with pytest.raises(pp_amd.PPError, match="no CPU fallback"):
        pp_amd._lib.Context(0)
    with pytest.raises(pp_amd.PPError):
        pillars.create_pillars(np.zeros((3, 4)), np.zeros((2, 2, 9)), np.zeros((2, 3)),
                               2, 2, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PillarVoxelizer(VoxelConfig())


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import,
    link, include or load it (comments may mention it)."""
    pkg = os.path.join(ROOT, "3d-object-detection_amd")
    bad = re.compile(r"^\s*(import\s+oracle|from\s+oracle|from\s+\.+oracle)|libpp_oracle|"
                     r"#include\s+\"pp_oracle|ppo_[a-z_]+\s*\(|dlopen", re.M)
    checked = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f"{f} reaches into oracle/"
                checked += 1
    assert checked >= 10


def test_argument_validation_surface():
    """Host-side checks that need no device (dtype / shape / writability)."""
    import numpy as np
    from pp_amd import pillars
    with pytest.raises(TypeError):
        pillars.create_pillars(np.zeros((3, 4)), np.zeros((2, 2, 9), np.float32), np.zeros((2, 3)),
                               2, 2, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    ro = np.zeros((2, 3))
    ro.flags.writeable = False
    with pytest.raises(TypeError):
        pillars.create_pillars(np.zeros((3, 4)), np.zeros((2, 2, 9)), ro, 2, 2, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    with pytest.raises(IndexError):
        pillars.create_pillars(np.zeros((3, 3)), np.zeros((2, 2, 9)), np.zeros((2, 3)),
                               2, 2, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    with pytest.raises(IndexError):
        pillars.create_pillars(np.zeros((3, 4)), np.zeros((2, 18)), np.zeros((2, 3)),
                               2, 2, 1, 1, 0, 0, 0, 4, 4, 4, 4)
    with pytest.raises(IndexError):
        pillars.make_ious(np.zeros((5, 4, 2)), np.zeros((2, 4, 2)), np.zeros((5, 3)), np.zeros((2, 3)),
                          np.zeros((4, 2)))
    assert pillars.create_pillars.__doc__.startswith("pillars") and pillars.make_ious.__doc__.startswith("ious")
    assert pillars.__doc__.startswith("point pillars data prep functions")   # pillars.cpp:431-433


def test_pybind11_module_surface_and_loud_failure_without_gpu():
    """The reference's own kind of binding: an extension module named `pillars` exporting
    exactly make_ious / create_pillars with the reference's doc strings
    (data/pillars.cpp:429-435), built from csrc/pillars_module.cpp on the C ABI.  Without a
    HIP device a call must fail loudly (no CPU fallback); argument checks come first."""
    import importlib.util
    import numpy as np
    import torch
    import pp_amd
    path = pp_amd._lib.build_pybind_module()
    spec = importlib.util.spec_from_file_location("pillars", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.__doc__ == "point pillars data prep functions"
    assert sorted(n for n in dir(mod) if not n.startswith("_")) == ["create_pillars", "make_ious"]
    assert mod.create_pillars.__doc__.strip().endswith("pillars")
    assert mod.make_ious.__doc__.strip().endswith("ious")
    T, I = np.zeros((2, 2, 9)), np.zeros((2, 3))
    with pytest.raises(TypeError):      # f32 output: the reference would lose the writes
        mod.create_pillars(np.zeros((3, 4)), T.astype(np.float32), I, 2, 2, 1., 1., 0., 0., 0., 4., 4., 4., 4)
    with pytest.raises(IndexError):     # wrong rank, pybind11's index_error
        mod.create_pillars(np.zeros((3, 4)), np.zeros((2, 9)), I, 2, 2, 1., 1., 0., 0., 0., 4., 4., 4., 4)
    with pytest.raises(TypeError):      # missing positional arguments
        mod.create_pillars(np.zeros((3, 4)), T, I)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            mod.create_pillars(np.zeros((3, 4)), T, I, 2, 2, 1., 1., 0., 0., 0., 4., 4., 4., 4)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            mod.make_ious(np.zeros((1, 4, 2)), np.zeros((1, 4, 2)), np.zeros((1, 3)), np.zeros((1, 3)),
                          np.zeros((1, 1)))


def test_device_entry_points_reject_null_and_bad_sizes_without_touching_a_device():
    """Every device entry point validates its arguments before any HIP call: with a NULL context
    or NULL tensors it returns PP_ERR_VALUE and leaves a message (runs on a CPU-only box)."""
    from pp_amd import _lib
    L = _lib.lib()
    V = _lib.PP_ERR_VALUE
    prm = _lib.make_voxel_params(4, 8, 1, 1, 0, 0, 0, 4, 4, 4, 4, 0)
    tp = _lib.TargetParams(0.6, 4.0, 9, 0)
    calls = {
        "pp_voxelize_dev": lambda: L.pp_voxelize_dev(None, None, None, 0, None, 1, ctypes.byref(prm), None, None, None),
        "pp_voxelize_pfn_dev": lambda: L.pp_voxelize_pfn_dev(None, None, None, 0, None, 1, ctypes.byref(prm), None, 64,
                                                              None, None, None),
        "pp_voxelize_pfn_canvas_dev": lambda: L.pp_voxelize_pfn_canvas_dev(None, None, None, 0, None, 1,
                                                                            ctypes.byref(prm), None, 64, None, 4, 4, 1,
                                                                            None, None),
        "pp_subtract_mean_dev": lambda: L.pp_subtract_mean_dev(None, None, None, 1, 16, None),
        "pp_pfn_dense_dev": lambda: L.pp_pfn_dense_dev(None, None, None, 1, 8, 4, None, 64, None),
        "pp_scatter_canvas_dev": lambda: L.pp_scatter_canvas_dev(None, None, None, None, 1, 64, 8, None, 4, 4, 1),
        "pp_pfn_train_stats_dev": lambda: L.pp_pfn_train_stats_dev(None, None, None, 1, 8, 4, None, 64, None),
        "pp_pfn_train_backward_dev": lambda: L.pp_pfn_train_backward_dev(None, None, None, 1, 8, 4, None, None, None,
                                                                          None, 64, None),
        "pp_assign_targets_dev": lambda: L.pp_assign_targets_dev(None, None, 10, None, None, None, None, 0, None, None,
                                                                  None, None, None, None, ctypes.byref(tp), None, None),
        "pp_assign_targets_grid_dev": lambda: L.pp_assign_targets_grid_dev(None, None, 2, 2, 0.5, 2, None, 0, None, None,
                                                                            None, None, None, None, ctypes.byref(tp),
                                                                            None, None),
        "pp_make_ious_dev": lambda: L.pp_make_ious_dev(None, None, None, None, 3, 4, None, None, 3, 2, None),
        "pp_ingest_dev": lambda: L.pp_ingest_dev(None, None, None, 4, 5, None, 1.0, None),
        "pp_bias_relu_bn_dev": lambda: L.pp_bias_relu_bn_dev(None, None, None, 1, 4, 16, None, None, 4, 0),
        "pp_bias_relu_bn_nhwc_dev": lambda: L.pp_bias_relu_bn_nhwc_dev(None, None, None, 16, 4, None, None, 4, 0),
        "pp_relu_bn_train_fwd_dev": lambda: L.pp_relu_bn_train_fwd_dev(None, None, None, None, 1, 4, 16, None, None,
                                                                        1e-5, 0.1, None, None, None, None, None),
        "pp_relu_bn_train_bwd_dev": lambda: L.pp_relu_bn_train_bwd_dev(None, None, None, None, None, 0, 1, 4, 16, None,
                                                                        None, None, None, None, None, None),
        "pp_ctx_set_timing": lambda: L.pp_ctx_set_timing(None, 4),
    }
    for name, call in calls.items():
        rc = call()
        assert rc == V, (name, rc)
        assert L.pp_last_error(), name


def test_host_thread_pool_selftest_without_a_device():
    """The pool of host threads behind pp_create_pillars_f64 / pp_make_ious_f64 (gather, scatter, zero fill) needs no
    device to be exercised: thousands of back-to-back jobs in both forms (blocking; start / wait with the caller free),
    every part of every job exactly once with the right (part, parts) pair, for 1..16 threads -- including the
    sleep / wake path (a pause between jobs longer than the workers' spin)."""
    import time
    import pp_amd
    L = pp_amd._lib.lib()
    for threads in (1, 2, 3, 8, 16):
        assert L.pp_host_pool_selftest(threads, 3000, 10007) == 0, threads
    assert L.pp_host_pool_selftest(4, 0, 5) == 0 and L.pp_host_pool_selftest(4, 7, 0) == 0
    assert L.pp_host_pool_selftest(0, 1, 1) == pp_amd._lib.PP_ERR_VALUE
    for _ in range(5):                      # a fresh pool each time, used after its workers have gone to sleep
        assert L.pp_host_pool_selftest(8, 20, 1000) == 0
        time.sleep(0.002)
