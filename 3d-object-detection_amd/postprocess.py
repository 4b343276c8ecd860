"""Device-resident inference post-processing.

Counterpart of the per-sample tail of ``evaluate()`` (/root/reference
evaluate.py:231-245): scores/classes, threshold, ``box_nms`` over the anchor
rectangles, the first 100 survivors decoded by ``make_pred_boxes`` and moved to car
space.  Returns plain arrays instead of lyft ``Box`` objects.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class Detector:
    def __init__(self, anchors, anchor_cfg, canvas_height, x_step, y_step, x_min, y_min,
                 pos_thresh=0.5, nms_thresh=0.1, max_out=100, num_classes=9, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("Detector needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.a_centers = torch.as_tensor(np.ascontiguousarray(anchors["centers"]), **f64)
        self.a_wlh = torch.as_tensor(np.ascontiguousarray(anchors["wlh"]), **f64)
        self.a_yaw = torch.as_tensor(np.ascontiguousarray(anchors["yaw"]), **f64)
        self.a_xy = torch.as_tensor(np.ascontiguousarray(anchors["xy"]), **f64)
        self.max_out = int(max_out)
        self._prm = _lib.DecodeParams(anchor_cfg.fm_height, anchor_cfg.fm_width, anchor_cfg.per_cell,
                                      int(num_classes), float(pos_thresh), float(nms_thresh),
                                      self.max_out, 0, float(canvas_height), float(x_step),
                                      float(y_step), float(x_min), float(y_min))
        self._ctx = _lib.Context(self.device.index)

    def __call__(self, cls, reg):
        """``cls [Ac*C,H,W]``, ``reg [Ac*8,H,W]`` float32: ONE sample of PPModel's output -- returns
        ``(boxes[max_out,9] f64, kept[max_out] i32, count[1] i32)`` device tensors: car-space
        x,y,z,w,l,h,yaw,score,class; rows beyond ``count`` are zero.  With a leading batch
        dimension (``[B,Ac*C,H,W]``, B > 1) every sample is decoded in the same three launches and
        the results carry the batch dimension: ``boxes[B,max_out,9]``, ``kept[B,max_out]``,
        ``count[B]`` (evaluate.py:231-245 loops over the samples)."""
        batched = cls.dim() == 4 and cls.shape[0] > 1
        if cls.dim() == 3:
            cls, reg = cls[None], reg[None]
        cls, reg = self._cell_strided(cls), self._cell_strided(reg)
        B = int(cls.shape[0])
        if reg.shape[0] != B:
            raise ValueError("cls and reg differ in batch size")
        boxes = torch.empty((B, self.max_out, 9), dtype=torch.float64, device=self.device)
        kept = torch.empty((B, self.max_out), dtype=torch.int32, device=self.device)
        count = torch.empty((B,), dtype=torch.int32, device=self.device)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        rc = _lib.lib().pp_decode_batch_dev(
            self._ctx.handle, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), B,
            vp(cls), vp(reg), cls.stride(0), cls.stride(1), cls.stride(3), reg.stride(0), reg.stride(1), reg.stride(3),
            vp(self.a_centers), vp(self.a_wlh), vp(self.a_yaw), vp(self.a_xy),
            ctypes.byref(self._prm), vp(boxes), vp(kept), vp(count))
        _lib.check(rc, "pp_decode_batch_dev")
        if batched:
            return boxes, kept, count
        return boxes[0], kept[0], count

    @staticmethod
    def _cell_strided(t):
        """``t[B,C,H,W]`` as it is when cell y*W+x sits at a fixed pitch (NCHW planes, channels-last
        tensors and channel slices of them -- PPModel's eval outputs), else an NCHW copy."""
        if t.dtype == torch.float32 and t.stride(2) == t.shape[3] * t.stride(3):
            return t
        return t.float().contiguous()
