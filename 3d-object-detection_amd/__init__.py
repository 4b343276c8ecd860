"""MI355X-native PointPillars hot path (pillar voxelizer + anchor-target assignment).

The directory name is not a Python identifier; import it through the root-level
``pp_amd`` shim (``import pp_amd``), which registers this package as ``pp_amd``.

Layout (only what the hot path needs):
  csrc/          HIP kernels + the C ABI (include/pp_hip.h)  -> libpp_hip.so
  _lib.py        ctypes binding, build(), error mapping
  pillars.py     drop-in for the reference's pybind11 module (create_pillars, make_ious)
  voxelizer.py   device-resident voxelizer (torch tensors)
  targets.py     device-resident anchor-target assignment
  boxes.py       anchors / box geometry as flat arrays
  model.py loss.py   PyTorch-ROCm counterpart of model/model.py, model/loss.py
  shard.py       one-process-per-GPU sweep sharding (torch.distributed / RCCL)
  synth.py       synthetic clouds and boxes
"""
from . import _lib  # noqa: F401
from ._lib import ORDER_ROW_MAJOR, ORDER_SCRAMBLED, PPError, build  # noqa: F401

__version__ = "0.1.0"
