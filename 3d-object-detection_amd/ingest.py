"""Device-resident lidar ingest.

Counterpart of the point preparation in ``PPDataset.__getitem__``
(/root/reference data/dataset.py:51-88): ``LidarPointCloud.from_file`` ->
``transform(transmat)`` -> ``remove_close(min_dist)`` -> ``np.hstack`` over
``num_sweeps`` sweeps.  The raw file rows are the only bytes that cross PCIe.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def transform_matrix(translation, rotation_matrix, inverse=False):
    """lyft_dataset_sdk.utils.geometry_utils.transform_matrix (absent, recalled):
    4x4 homogeneous matrix from a translation and a 3x3 rotation; ``inverse``
    gives the inverse rigid transform."""
    tm = np.eye(4)
    rot = np.asarray(rotation_matrix, np.float64)
    trans = np.asarray(translation, np.float64)
    if inverse:
        tm[:3, :3] = rot.T
        tm[:3, 3] = rot.T.dot(-trans)
    else:
        tm[:3, :3] = rot
        tm[:3, 3] = trans
    return tm


class LidarIngest:
    def __init__(self, device=None, min_dist=0.001):   # dataset.py:32
        if not torch.cuda.is_available():
            raise RuntimeError("LidarIngest needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self.min_dist = float(min_dist)
        self._ctx = _lib.Context(self.device.index)

    def __call__(self, sweeps):
        """``sweeps``: list of ``(raw[n_s, C>=4] float32 tensor or ndarray, transmat 4x4)``.
        Returns the aggregated ``[sum n_s, 4]`` f32 device tensor (removed points
        have x = NaN and are ignored by the voxelizer)."""
        raws = []
        for raw, _ in sweeps:
            r = torch.as_tensor(raw, dtype=torch.float32)
            if r.dim() != 2 or r.shape[1] < 4:
                raise ValueError("raw sweep must be [n, >=4] float32")
            raws.append(r.to(self.device, non_blocking=True).contiguous())
        total = sum(int(r.shape[0]) for r in raws)
        out = torch.empty((total, 4), dtype=torch.float32, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        cols = {int(r.shape[1]) for r in raws}
        off = 0
        # sweeps with the same row width go sixteen at a time in one launch
        for lo in range(0, len(raws), _lib.MAX_INGEST_SWEEPS if len(cols) == 1 else 1):
            grp = raws[lo:lo + (_lib.MAX_INGEST_SWEEPS if len(cols) == 1 else 1)]
            k = len(grp)
            ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in grp])
            ns = (ctypes.c_int64 * k)(*[int(r.shape[0]) for r in grp])
            mats = np.ascontiguousarray(np.stack([np.asarray(m, np.float64).reshape(4, 4)
                                                  for _, m in sweeps[lo:lo + k]]))
            rc = _lib.lib().pp_ingest_sweeps_dev(
                self._ctx.handle, stream, k, ptrs, ns, int(grp[0].shape[1]),
                mats.ctypes.data_as(ctypes.c_void_p), self.min_dist, ctypes.c_void_p(out.data_ptr() + off * 16))
            _lib.check(rc, "pp_ingest_sweeps_dev")
            off += sum(int(r.shape[0]) for r in grp)
        return out
