"""Device-resident anchor-target assignment and IoU (torch tensors in/out).

Counterpart of ``create_target`` (/root/reference utils/box_utils.py:162-232)
with ``make_ious`` (data/pillars.cpp:400-427) and ``make_target``
(box_utils.py:70-109) fused into HIP kernels; outputs are the float32 tensors
``PPDataset.__getitem__`` returns (data/dataset.py:117-118).
"""
import ctypes

import numpy as np
import torch

from . import _lib, boxes


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


class TargetAssigner:
    """``assign(gt...)`` returns ``(cls_targets[A,C] f32, reg_targets[A,9] f32)`` for one
    sample.  ``anchors`` is either the dict of flat arrays of ``boxes.make_anchors`` (held
    on the device: the counterpart of anchor_boxes.pkl, train_prep.py:115-120) or a
    ``boxes.AnchorConfig`` -- then the kernels evaluate the anchor grid on the fly from a
    [per_cell,13] table and no per-anchor array exists at all (same results, bit for bit)."""

    def __init__(self, anchors, canvas_height, pos_thresh=0.6, num_classes=9, device=None, lib_=None):
        """``lib_``: another build of the library (``_lib.variant_lib``), for A/B tests; default the product's."""
        self._L = lib_ if lib_ is not None else _lib.lib()
        if isinstance(anchors, boxes.AnchorConfig):
            self._init_common(canvas_height, pos_thresh, num_classes, device)
            self.grid = anchors
            self.A = anchors.num_anchors
            self.types = torch.as_tensor(boxes.anchor_type_table(anchors), dtype=torch.float64,
                                         device=self.device).contiguous()
            self.a_corners = self.a_centers = self.a_wlh = self.a_yaw = None
            return
        self._init_common(canvas_height, pos_thresh, num_classes, device)
        self.grid = None
        f64 = dict(dtype=torch.float64, device=self.device)
        self.a_corners = torch.as_tensor(np.ascontiguousarray(anchors["corners"]), **f64).contiguous()
        self.a_centers = torch.as_tensor(np.ascontiguousarray(anchors["centers"]), **f64).contiguous()
        self.a_wlh = torch.as_tensor(np.ascontiguousarray(anchors["wlh"]), **f64).contiguous()
        self.a_yaw = torch.as_tensor(np.ascontiguousarray(anchors["yaw"]), **f64).contiguous()
        self.A = self.a_corners.shape[0]

    def _init_common(self, canvas_height, pos_thresh, num_classes, device):
        if not torch.cuda.is_available():
            raise RuntimeError("TargetAssigner needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self.canvas_height = float(canvas_height)
        self.pos_thresh = float(pos_thresh)
        self.num_classes = int(num_classes)
        self._ctx = _lib.Context(self.device.index, lib_=self._L)
        self._prm = _lib.TargetParams(self.pos_thresh, self.canvas_height, self.num_classes, 0)

    def _check(self, rc, what):
        if rc != _lib.PP_OK:
            _lib.check(rc, what, self._L.pp_last_error())

    def _gt_to_device(self, gt_centers, gt_wlh, gt_yaw, gt_classes):
        gt_centers = np.asarray(gt_centers, np.float64).reshape(-1, 3)
        gt_wlh = np.asarray(gt_wlh, np.float64).reshape(-1, 3)
        gt_yaw = np.asarray(gt_yaw, np.float64).reshape(-1)
        centers_img, corners_img = boxes.boxes_to_image_space(gt_centers, gt_wlh, gt_yaw,
                                                              self.canvas_height)
        f64 = dict(dtype=torch.float64, device=self.device)
        return (torch.as_tensor(np.ascontiguousarray(corners_img), **f64),
                torch.as_tensor(np.ascontiguousarray(centers_img), **f64),
                torch.as_tensor(np.ascontiguousarray(gt_centers), **f64),
                torch.as_tensor(np.ascontiguousarray(gt_wlh), **f64),
                torch.as_tensor(np.ascontiguousarray(gt_yaw), **f64),
                torch.as_tensor(np.asarray(gt_classes, np.int32).reshape(-1), dtype=torch.int32,
                                device=self.device))

    def assign(self, gt_centers, gt_wlh, gt_yaw, gt_classes, check=False):
        """gt_* describe the sample's boxes in canvas space (the fields of the lyft
        ``Box`` objects the reference pickles: center, wlh, yaw, class index)."""
        g = self._gt_to_device(gt_centers, gt_wlh, gt_yaw, gt_classes)
        return self.assign_device(*g, check=check)

    def assign_device(self, g_corners, g_centers_img, g_centers, g_wlh, g_yaw, g_class, check=False, out=None):
        """``out``: an optional ``(cls[A,C], reg[A,9])`` pair of contiguous f32 tensors to fill (a loop that re-uses its
        outputs saves two allocator calls per sample -- a third of a call's host time at one sample per launch)."""
        G = int(g_corners.shape[0])
        if out is None:
            cls_t = torch.empty((self.A, self.num_classes), dtype=torch.float32, device=self.device)
            reg_t = torch.empty((self.A, 9), dtype=torch.float32, device=self.device)
        else:
            cls_t, reg_t = out
            if cls_t.shape != (self.A, self.num_classes) or reg_t.shape != (self.A, 9) or \
                    cls_t.dtype != torch.float32 or reg_t.dtype != torch.float32 or \
                    not (cls_t.is_contiguous() and reg_t.is_contiguous()) or cls_t.device != self.device:
                raise ValueError("out: contiguous f32 (cls[A,C], reg[A,9]) on " + str(self.device))
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.grid is not None:
            c = self.grid
            rc = self._L.pp_assign_targets_grid_dev(
                self._ctx.handle, stream, c.fm_height, c.fm_width, float(c.fm_scale), c.per_cell,
                _vp(self.types), G, _vp(g_corners), _vp(g_centers_img), _vp(g_centers), _vp(g_wlh),
                _vp(g_yaw), _vp(g_class), ctypes.byref(self._prm), _vp(cls_t), _vp(reg_t))
            self._check(rc, "pp_assign_targets_grid_dev")
        else:
            rc = self._L.pp_assign_targets_dev(
                self._ctx.handle, stream, self.A, _vp(self.a_corners), _vp(self.a_centers),
                _vp(self.a_wlh), _vp(self.a_yaw), G, _vp(g_corners), _vp(g_centers_img),
                _vp(g_centers), _vp(g_wlh), _vp(g_yaw), _vp(g_class), ctypes.byref(self._prm),
                _vp(cls_t), _vp(reg_t))
            self._check(rc, "pp_assign_targets_dev")
        if check:
            self._check(self._L.pp_iou_check(self._ctx.handle, stream), "pp_assign_targets_dev")
        return cls_t, reg_t

    # ------------------------------------------------------------------ a batch of samples per launch
    def upload_batch(self, gts):
        """``gts``: one dict per sample (centers / wlh / yaw / classes, canvas space).  Returns the
        device hand-over ``assign_batch_device`` consumes: ``(g_counts, packed)`` -- the samples' boxes
        concatenated in sample order in ONE f64 device buffer (one host-to-device copy per step instead
        of six per sample), laid out [corners 8 | centers_img 3 | centers 3 | wlh 3 | yaw 1] per array,
        then the int32 classes."""
        counts = [int(np.asarray(g["yaw"]).reshape(-1).shape[0]) for g in gts]
        if not 1 <= len(counts) <= _lib.MAX_BATCH:
            raise ValueError(f"a batch is 1..{_lib.MAX_BATCH} samples, got {len(counts)}")
        T = sum(counts)
        host = np.zeros(max(T, 1) * 19, np.float64)   # 18 f64 columns + the classes' int32 (padded to 8 B)
        corners, cimg = host[:T * 8].reshape(T, 4, 2), host[T * 8:T * 11].reshape(T, 3)
        cen, wlh = host[T * 11:T * 14].reshape(T, 3), host[T * 14:T * 17].reshape(T, 3)
        yaw, cls = host[T * 17:T * 18], host[T * 18:].view(np.int32)[:T]
        o = 0
        for g, n in zip(gts, counts):
            if n:
                c = np.asarray(g["centers"], np.float64).reshape(n, 3)
                w = np.asarray(g["wlh"], np.float64).reshape(n, 3)
                y = np.asarray(g["yaw"], np.float64).reshape(n)
                ci, ki = boxes.boxes_to_image_space(c, w, y, self.canvas_height)
                corners[o:o + n], cimg[o:o + n], cen[o:o + n], wlh[o:o + n], yaw[o:o + n] = ki, ci, c, w, y
                cls[o:o + n] = np.asarray(g["classes"], np.int32).reshape(n)
            o += n
        return counts, torch.from_numpy(host).to(self.device, non_blocking=False)

    def assign_batch(self, gts, check=False):
        """``create_target`` for every sample of a step in ONE launch (data/dataset.py:113-118 runs it per
        sample; config.py:135: four samples per step).  Returns ``(cls_targets[B,A,C], reg_targets[B,A,9])``."""
        return self.assign_batch_device(*self.upload_batch(gts), check=check)

    def assign_batch_device(self, g_counts, packed, check=False, out=None):
        B, T = len(g_counts), int(sum(g_counts))
        if packed.dtype != torch.float64 or packed.numel() < max(T, 1) * 19 or packed.device != self.device:
            raise ValueError("packed ground truths: the f64 device buffer of upload_batch()")
        if out is None:
            out = (torch.empty((B, self.A, self.num_classes), dtype=torch.float32, device=self.device),
                   torch.empty((B, self.A, 9), dtype=torch.float32, device=self.device))
        cls_t, reg_t = out
        if cls_t.shape != (B, self.A, self.num_classes) or reg_t.shape != (B, self.A, 9) or \
                cls_t.dtype != torch.float32 or reg_t.dtype != torch.float32 or \
                not (cls_t.is_contiguous() and reg_t.is_contiguous()):
            raise ValueError("out: contiguous f32 (cls[B,A,C], reg[B,A,9])")
        base = packed.data_ptr()
        gp = [ctypes.c_void_p(base + 8 * T * k) if T else None for k in (0, 8, 11, 14, 17, 18)]
        counts = (ctypes.c_int32 * B)(*g_counts)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.grid is not None:
            c = self.grid
            rc = self._L.pp_assign_targets_grid_batch_dev(
                self._ctx.handle, stream, B, counts, c.fm_height, c.fm_width, float(c.fm_scale), c.per_cell,
                _vp(self.types), *gp, ctypes.byref(self._prm), _vp(cls_t), _vp(reg_t))
        else:
            rc = self._L.pp_assign_targets_batch_dev(
                self._ctx.handle, stream, B, counts, self.A, _vp(self.a_corners), _vp(self.a_centers),
                _vp(self.a_wlh), _vp(self.a_yaw), *gp, ctypes.byref(self._prm), _vp(cls_t), _vp(reg_t))
        self._check(rc, "pp_assign_targets_batch_dev")
        if check:
            self._check(self._L.pp_iou_check(self._ctx.handle, stream), "pp_assign_targets_batch_dev")
        return cls_t, reg_t

    def ious(self, g_corners_img, g_centers_img, check=True):
        """Dense [A,G] f64 IoU matrix on the device (make_ious, pillars.cpp:400-427)."""
        if self.grid is not None:
            raise RuntimeError("the dense IoU matrix needs the anchor arrays: build the assigner from "
                               "boxes.make_anchors(cfg)")
        f64 = dict(dtype=torch.float64, device=self.device)
        gc = torch.as_tensor(np.ascontiguousarray(g_corners_img), **f64).contiguous()
        gn = torch.as_tensor(np.ascontiguousarray(g_centers_img), **f64).contiguous()
        G = int(gc.shape[0])
        out = torch.empty((self.A, G), **f64)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._L.pp_make_ious_dev(self._ctx.handle, stream, _vp(self.a_corners),
                                         _vp(self.a_centers), 3, self.A, _vp(gc), _vp(gn),
                                         int(gn.shape[1]) if G else 3, G, _vp(out))
        self._check(rc, "pp_make_ious_dev")
        if check:
            self._check(self._L.pp_iou_check(self._ctx.handle, stream), "pp_make_ious_dev")
        return out
