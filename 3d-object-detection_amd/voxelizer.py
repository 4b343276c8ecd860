"""Device-resident pillar voxelizer (torch tensors in, torch tensors out).

Counterpart of the voxel stage of ``PPDataset.__getitem__`` (/root/reference
data/dataset.py:88-106): ``np.zeros`` + ``pillars.create_pillars`` + transpose
to ``[9,P,N]`` + ``.float()`` + ``indices.long()``, done by the HIP kernels of
libpp_hip.so on the caller's current HIP stream.  torch is used for device
memory and streams only.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib


@dataclass(frozen=True)
class VoxelConfig:
    """The grid constants of config.py:46-61,119-120 as explicit parameters."""
    max_points_per_pillar: int = 100     # N  (config.py:119 ships 200)
    max_pillars: int = 12000             # P  (config.py:120 ships 24000)
    x_step: float = 0.2
    y_step: float = 0.2
    x_min: float = -50.0
    y_min: float = -50.0
    z_min: float = -10.0
    x_max: float = 50.0
    y_max: float = 50.0
    z_max: float = 10.0
    canvas_height: int = 500
    # Pillar order.  The reference emits pillars in boost::unordered_map iteration order
    # (pillars.cpp:335) and, with more occupied cells than max_pillars, drops whatever iterates
    # last: a spatially scattered subset.  ORDER_SCRAMBLED is the deterministic stand-in for that
    # (default); ORDER_ROW_MAJOR drops the rows with the largest canvas_y as one region.
    order: int = _lib.ORDER_SCRAMBLED

    @staticmethod
    def reference_default():
        """config.py:46-53,60,119-120 exactly."""
        return VoxelConfig(200, 24000, .2, .2, -60, -60, -10, 60, 60, 10, 600)

    @staticmethod
    def square(half, step, max_pillars, max_points, z_min=-10.0, z_max=10.0,
               order=_lib.ORDER_SCRAMBLED):
        n = int(round(2 * half / step))
        return VoxelConfig(max_points, max_pillars, step, step, -half, -half, z_min,
                           half, half, z_max, n, order)

    @staticmethod
    def rect(x_range, y_range, x_step, y_step, max_pillars, max_points, z_range=(-10.0, 10.0),
             canvas_height=None, order=_lib.ORDER_SCRAMBLED):
        """Any grid create_pillars accepts (pillars.cpp:236-249 takes nine independent scalars): different steps
        and ranges per axis, a range that is not centred, and a ``canvas_height`` of the caller's choosing -- the
        reference just computes ``(canvas_height - 1) - floor((y - y_min) / y_step)`` (pillars.cpp:278-280), rows
        beyond the canvas or negative ones included.  Default ``canvas_height``: the grid's row count."""
        if canvas_height is None:
            canvas_height = int(round((y_range[1] - y_range[0]) / y_step))
        return VoxelConfig(max_points, max_pillars, x_step, y_step, x_range[0], y_range[0], z_range[0],
                           x_range[1], y_range[1], z_range[1], canvas_height, order)

    def grid_args(self):
        """The nine grid scalars in create_pillars' positional order (pillars.cpp:241-249)."""
        return (self.x_step, self.y_step, self.x_min, self.y_min, self.z_min, self.x_max, self.y_max, self.z_max,
                self.canvas_height)

    @property
    def canvas_width(self):
        # config.py:61
        return int((self.x_max - self.x_min) / self.x_step)

    def params(self):
        return _lib.make_voxel_params(self.max_points_per_pillar, self.max_pillars, self.x_step,
                                      self.y_step, self.x_min, self.y_min, self.z_min, self.x_max,
                                      self.y_max, self.z_max, self.canvas_height, self.order)

    def algorithmic_bytes(self, n_points):
        """mandatory input read + mandatory dense write per sweep."""
        return (16 * n_points + 4 * _lib.NUM_FEATURES * self.max_pillars * self.max_points_per_pillar
                + 24 * self.max_pillars)


class PillarVoxelizer:
    """``voxelizer(points) -> (pillars[B,9,P,N] f32, indices[B,P,3] i64)``.

    ``points`` is a float32 CUDA(HIP) tensor ``[B, n_cap, 4]`` (or ``[n,4]`` for a
    single sweep); ``n_points`` gives the valid row count per sweep (default:
    all rows).  Everything runs on ``torch.cuda.current_stream()``.
    """

    def __init__(self, cfg: VoxelConfig, device=None, data_mean=None):
        if not torch.cuda.is_available():
            raise RuntimeError("PillarVoxelizer needs a HIP device; there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self._ctx = _lib.Context(self.device.index)
        self._prm = cfg.params()
        self.data_mean = None
        self.set_data_mean(data_mean)

    def set_data_mean(self, data_mean):
        """The optional dataset mean of the pillar tensor (pillar_means.pkl: a flat
        9*P*N float32 tensor, data/dataset.py:102-105, make_means.py); it is subtracted from
        every sweep's ``[9,P,N]`` output.  ``None`` switches it off."""
        if data_mean is None:
            self.data_mean = None
            return
        n = _lib.NUM_FEATURES * self.cfg.max_pillars * self.cfg.max_points_per_pillar
        m = torch.as_tensor(data_mean, dtype=torch.float32).reshape(-1)
        if m.numel() != n:
            raise ValueError(f"data_mean has {m.numel()} elements, the pillar tensor {n}")
        self.data_mean = m.to(self.device).contiguous()

    def reserve(self, batch, max_points):
        _lib.check(_lib.lib().pp_voxelize_reserve(self._ctx.handle, int(batch), int(max_points),
                                                  ctypes.byref(self._prm)), "pp_voxelize_reserve")

    def set_timing(self, slots):
        _lib.check(_lib.lib().pp_ctx_set_timing(self._ctx.handle, int(slots)), "pp_ctx_set_timing")

    def read_kernel_ms(self, which, cap=4096):
        """Durations (ms, oldest first) of kernel ``which`` (_lib.KERNEL_SPLIT / _TILE / _EMIT)
        of the calls since ``set_timing``; reading KERNEL_EMIT empties the ring."""
        buf = (ctypes.c_float * cap)()
        cnt = ctypes.c_int()
        _lib.check(_lib.lib().pp_ctx_read_kernel_ms(self._ctx.handle, int(which), buf, cap, ctypes.byref(cnt)),
                   "pp_ctx_read_kernel_ms")
        return [buf[i] for i in range(cnt.value)]

    def read_emit_ms(self, cap=4096):
        return self.read_kernel_ms(_lib.KERNEL_EMIT, cap)

    def check(self):
        """Synchronises the current stream and raises if a launch on it failed (the device entry
        points never synchronise by themselves): pp_voxelize_check."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().pp_voxelize_check(self._ctx.handle, ctypes.c_void_p(stream)), "pp_voxelize_check")

    def _prep(self, points, n_points):
        if points.dim() == 2:
            points = points.unsqueeze(0)
        if (points.dim() != 3 or points.shape[-1] != 4 or points.dtype != torch.float32
                or not points.is_cuda or points.device != self.device):
            raise ValueError("points must be a float32 tensor [B, n, 4] on " + str(self.device))
        if not points.is_contiguous():
            points = points.contiguous()
        B, ncap = points.shape[0], points.shape[1]
        if n_points is None:
            n_points = [ncap] * B
        return points, B, ncap, (ctypes.c_int32 * B)(*[int(v) for v in n_points])

    def pfn(self, points, pfn_params, n_points=None, out=None, return_counts=False):
        """Voxelizer with the feature net fused in (inference): returns
        ``(features[B,64,P] f32, indices[B,P,3] i64)`` -- PPFeatureNet's output
        (model/model.py:31-40) without ever building the dense [9,P,N] tensor.
        ``pfn_params`` is the [64,12] tensor of ``PPFeatureNet.fused_params()``."""
        points, B, ncap, n_arr = self._prep(points, n_points)
        P = self.cfg.max_pillars
        self._no_mean("pfn")
        if (pfn_params.shape != (64, 12) or pfn_params.dtype != torch.float32
                or pfn_params.device != self.device or not pfn_params.is_contiguous()):
            raise ValueError("pfn_params must be a contiguous float32 [64,12] tensor on " + str(self.device))
        if out is None:
            feats = torch.empty((B, 64, P), dtype=torch.float32, device=self.device)
            indices = torch.empty((B, P, 3), dtype=torch.int64, device=self.device)
        else:
            feats, indices = out
        counts = (torch.empty((B, 2), dtype=torch.int32, device=self.device)
                  if return_counts else None)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = _lib.lib().pp_voxelize_pfn_dev(
            self._ctx.handle, ctypes.c_void_p(stream), ctypes.c_void_p(points.data_ptr()),
            ncap, n_arr, B, ctypes.byref(self._prm), ctypes.c_void_p(pfn_params.data_ptr()), 64,
            ctypes.c_void_p(feats.data_ptr()), ctypes.c_void_p(indices.data_ptr()),
            ctypes.c_void_p(counts.data_ptr()) if counts is not None else None)
        _lib.check(rc, "pp_voxelize_pfn_dev")
        if return_counts:
            return feats, indices, counts
        return feats, indices

    def pfn_canvas(self, points, pfn_params, canvas_hw, n_points=None, channels_last=True,
                   out=None, return_counts=False, reuse=False):
        """Voxelizer + feature net + PPScatter (model/model.py:31-62) in one pass:
        returns ``(canvas[B,64,H,W] f32, indices[B,P,3] i64)``; with
        ``channels_last`` the canvas is a channels-last tensor (same logical shape,
        memory [B,H,W,64]) that MIOpen's NHWC convolutions consume directly.

        ``reuse=True`` (with ``out``): the caller promises that ``out`` is exactly what the previous
        ``pfn_canvas`` call with these buffers returned (same batch, canvas untouched since) -- then only
        the pixels that call wrote are zeroed instead of the whole canvas
        (pp_voxelize_pfn_canvas_reuse_dev)."""
        points, B, ncap, n_arr = self._prep(points, n_points)
        P = self.cfg.max_pillars
        self._no_mean("pfn_canvas")
        H, W = int(canvas_hw[0]), int(canvas_hw[1])
        if (pfn_params.shape != (64, 12) or pfn_params.dtype != torch.float32
                or pfn_params.device != self.device or not pfn_params.is_contiguous()):
            raise ValueError("pfn_params must be a contiguous float32 [64,12] tensor on " + str(self.device))
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        if out is None:
            canvas = torch.empty((B, 64, H, W), dtype=torch.float32, device=self.device, memory_format=fmt)
            indices = torch.empty((B, P, 3), dtype=torch.int64, device=self.device)
        else:
            canvas, indices = out
            if (canvas.shape != (B, 64, H, W) or canvas.dtype != torch.float32
                    or not canvas.is_contiguous(memory_format=fmt)
                    or indices.shape != (B, P, 3) or indices.dtype != torch.int64
                    or not indices.is_contiguous()):
                raise ValueError("out buffers have the wrong shape/dtype/layout")
        counts = (torch.empty((B, 2), dtype=torch.int32, device=self.device)
                  if return_counts else None)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = _lib.lib().pp_voxelize_pfn_canvas_reuse_dev(
            self._ctx.handle, ctypes.c_void_p(stream), ctypes.c_void_p(points.data_ptr()),
            ncap, n_arr, B, ctypes.byref(self._prm), ctypes.c_void_p(pfn_params.data_ptr()), 64,
            ctypes.c_void_p(canvas.data_ptr()), H, W, 1 if channels_last else 0,
            ctypes.c_void_p(indices.data_ptr()),
            ctypes.c_void_p(counts.data_ptr()) if counts is not None else None,
            ctypes.c_void_p(indices.data_ptr()) if (reuse and out is not None) else None)
        _lib.check(rc, "pp_voxelize_pfn_canvas_reuse_dev")
        if return_counts:
            return canvas, indices, counts
        return canvas, indices

    def _no_mean(self, what):
        if self.data_mean is not None:
            raise RuntimeError(f"{what}: the fused feature net skips the zero-padded slots, which a data_mean "
                               "makes non-zero; use the dense path (voxelizer(points) + PPFeatureNet)")

    # -- software-pipelined mode ------------------------------------------------------------
    LAG = 3   # calls between handing a batch in and getting it back

    def submit(self, points, n_points=None, out=None, return_counts=False):
        """One step of the software pipeline (pp_voxelize_step_dev): ONE launch runs the split stage
        of ``points``, the tile stage of the batch submitted one call ago, the order stage of the one
        before that and the emit stage (the dense store) of the batch submitted THREE calls ago, side
        by side -- the way the reference's DataLoader workers prepare the next batches while the model
        runs the current one (train.py:120-121).

        Returns the ``(pillars, indices[, counts])`` of the batch submitted three calls ago
        (bit-identical to ``__call__`` on that batch), or ``None`` while the pipeline fills.
        ``points=None`` drains (``LAG`` such calls flush everything).  ``out`` buffers (optional)
        receive that older batch.  Everything runs on the current stream.  A refused call (``ValueError``: wrong
        stream, bad argument) leaves the pipeline as it was; a failed launch resets it (``_submit_failed``)."""
        cfg = self.cfg
        P, N = cfg.max_pillars, cfg.max_points_per_pillar
        inflight = getattr(self, "_inflight", None)
        if inflight is None:
            inflight = self._inflight = [None, None, None]   # [split done, tiled, ordered]: batch sizes
        due = inflight[2]
        pillars = indices = counts = None
        if due is not None:
            if out is None:
                pillars = torch.empty((due, _lib.NUM_FEATURES, P, N), dtype=torch.float32, device=self.device)
                indices = torch.empty((due, P, 3), dtype=torch.int64, device=self.device)
            else:
                pillars, indices = out
                if (pillars.shape != (due, _lib.NUM_FEATURES, P, N) or pillars.dtype != torch.float32
                        or indices.shape != (due, P, 3) or indices.dtype != torch.int64
                        or not pillars.is_contiguous() or not indices.is_contiguous()):
                    raise ValueError("out buffers have the wrong shape/dtype/layout for the batch that is due")
            if return_counts:
                counts = torch.empty((due, 2), dtype=torch.int32, device=self.device)
        nxt = self._prep(points, n_points) if points is not None else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        emitted = ctypes.c_int(0)
        vp = ctypes.c_void_p
        rc = _lib.lib().pp_voxelize_step_dev(
            self._ctx.handle, vp(stream), vp(nxt[0].data_ptr()) if nxt else None,
            nxt[2] if nxt else 0, nxt[3] if nxt else None, nxt[1] if nxt else 0, ctypes.byref(self._prm),
            vp(pillars.data_ptr()) if pillars is not None else None,
            vp(indices.data_ptr()) if indices is not None else None,
            vp(counts.data_ptr()) if counts is not None else None, ctypes.byref(emitted))
        if rc != _lib.PP_OK:
            self._submit_failed(rc, "pp_voxelize_step_dev")
        self._inflight = [nxt[1] if nxt else None, inflight[0], inflight[1]]
        if due is None:
            return None
        assert emitted.value == 1
        if self.data_mean is not None:       # dataset.py:102-105
            rc = _lib.lib().pp_subtract_mean_dev(
                self._ctx.handle, vp(stream), vp(pillars.data_ptr()), due,
                _lib.NUM_FEATURES * P * N, vp(self.data_mean.data_ptr()))
            _lib.check(rc, "pp_subtract_mean_dev")
        return (pillars, indices, counts) if return_counts else (pillars, indices)

    def submit_pfn_canvas(self, points, pfn_params, canvas_hw, n_points=None, channels_last=True,
                          return_counts=False):
        """``submit`` with the emit stage as the fused feature net + scatter (pp_voxelize_step_pfn_canvas_dev;
        PPFeatureNet.forward and PPScatter.forward, model/model.py:31-62): ONE launch per call; the batch
        submitted ``LAG`` calls ago comes back as ``(canvas[B,64,H,W] f32, indices[B,P,3] i64[, counts])`` -- the
        result of ``pfn_canvas`` on that batch, bit for bit -- or ``None`` while the pipeline fills;
        ``points=None`` drains.  The batches in flight are shared with ``submit`` (same pipeline; the call
        that is made when a batch is due decides which form it comes out in).

        Two canvases are used in turn: a call fills one and, in the same launch, zeroes the pixels the previous
        call wrote into the other.  THE RETURNED CANVAS IS THEREFORE VALID UNTIL THE NEXT ``submit_pfn_canvas``
        CALL (in stream order: work enqueued before that call reads it safely); clone it to keep it."""
        cfg = self.cfg
        P = cfg.max_pillars
        self._no_mean("submit_pfn_canvas")
        H, W = int(canvas_hw[0]), int(canvas_hw[1])
        if (pfn_params.shape != (64, 12) or pfn_params.dtype != torch.float32
                or pfn_params.device != self.device or not pfn_params.is_contiguous()):
            raise ValueError("pfn_params must be a contiguous float32 [64,12] tensor on " + str(self.device))
        inflight = getattr(self, "_inflight", None)
        if inflight is None:
            inflight = self._inflight = [None, None, None]
        due = inflight[2]
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        st = getattr(self, "_pc", None)
        key = (due, H, W, bool(channels_last))
        if due is not None and (st is None or st["key"] != key):
            st = self._pc = {"key": key, "turn": 0, "filled": [0, 0],
                             "canvas": [torch.zeros((due, 64, H, W), dtype=torch.float32, device=self.device)
                                        .contiguous(memory_format=fmt) for _ in range(2)],
                             "indices": [torch.empty((due, P, 3), dtype=torch.int64, device=self.device)
                                         for _ in range(2)]}
        vp = ctypes.c_void_p
        canvas = indices = counts = clear_c = clear_i = None
        clear_b = 0
        if due is not None:
            k = st["turn"]
            if st["filled"][k]:                      # (a canvas that could not be cleared by a launch: rare)
                st["canvas"][k].zero_()
                st["filled"][k] = 0
            canvas, indices = st["canvas"][k], st["indices"][k]
            if st["filled"][1 - k]:
                clear_c, clear_i, clear_b = st["canvas"][1 - k], st["indices"][1 - k], st["filled"][1 - k]
            if return_counts:
                counts = torch.empty((due, 2), dtype=torch.int32, device=self.device)
        nxt = self._prep(points, n_points) if points is not None else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        emitted = ctypes.c_int(0)
        rc = _lib.lib().pp_voxelize_step_pfn_canvas_dev(
            self._ctx.handle, vp(stream), vp(nxt[0].data_ptr()) if nxt else None,
            nxt[2] if nxt else 0, nxt[3] if nxt else None, nxt[1] if nxt else 0, ctypes.byref(self._prm),
            vp(pfn_params.data_ptr()), 64, vp(canvas.data_ptr()) if canvas is not None else None, H, W,
            1 if channels_last else 0, vp(indices.data_ptr()) if indices is not None else None,
            vp(counts.data_ptr()) if counts is not None else None,
            vp(clear_c.data_ptr()) if clear_c is not None else None,
            vp(clear_i.data_ptr()) if clear_i is not None else None, clear_b, ctypes.byref(emitted))
        if rc != _lib.PP_OK:                        # same rule as submit()
            if rc != _lib.PP_ERR_VALUE:
                self._pc = None                      # the canvases' state is unknown: start from zeroed ones
            self._submit_failed(rc, "pp_voxelize_step_pfn_canvas_dev")
        self._inflight = [nxt[1] if nxt else None, inflight[0], inflight[1]]
        if due is None:
            return None
        assert emitted.value == 1
        st["filled"][k] = due
        if clear_c is not None:
            st["filled"][1 - k] = 0
        st["turn"] = 1 - k
        return (canvas, indices, counts) if return_counts else (canvas, indices)

    def _submit_failed(self, rc, what):
        """The contract of include/pp_hip.h, on both sides alike.  A call REFUSED before anything was launched
        (``PP_ERR_VALUE``: another stream while batches are in flight, a bad argument, missing output buffers for
        the batch that is due) changes nothing: the batches in flight stay in flight, here and in the library, and
        the next valid call carries on (a stray call does not cost three good batches).  Any other failure (a
        failed launch or allocation) abandons the batches in flight on both sides -- their results are never
        returned and the next submit starts an empty pipeline."""
        msg = _lib.lib().pp_last_error()
        if rc != _lib.PP_ERR_VALUE:
            _lib.lib().pp_voxelize_step_reset(self._ctx.handle)
            self._inflight = [None, None, None]
            what += " (pipeline reset)"
        _lib.check(rc, what, msg)

    def step_kernel_name(self, batch):
        """The k_step instance ``submit`` emits a batch of ``batch`` sweeps with, as a kernel trace prints it."""
        buf = ctypes.create_string_buffer(64)
        _lib.check(_lib.lib().pp_voxelize_step_kernel_name(ctypes.byref(self._prm), int(batch), buf, 64),
                   "pp_voxelize_step_kernel_name")
        return buf.value.decode()

    def reset_stream(self):
        """Forgets the batches in flight in ``submit``'s pipeline (their results are never returned)."""
        _lib.check(_lib.lib().pp_voxelize_step_reset(self._ctx.handle), "pp_voxelize_step_reset")
        self._inflight = [None, None, None]

    def stream(self, batches, n_points=None):
        """Generator over an iterable of point tensors: yields ``(pillars, indices)`` per batch, in
        order, from the software pipeline of ``submit``."""
        for pts in batches:
            r = self.submit(pts, n_points)
            if r is not None:
                yield r
        for _ in range(self.LAG):
            r = self.submit(None)
            if r is not None:
                yield r

    def __call__(self, points, n_points=None, out=None, return_counts=False):
        cfg = self.cfg
        points, B, ncap, n_arr = self._prep(points, n_points)
        P, N = cfg.max_pillars, cfg.max_points_per_pillar
        if out is None:
            pillars = torch.empty((B, _lib.NUM_FEATURES, P, N), dtype=torch.float32, device=self.device)
            indices = torch.empty((B, P, 3), dtype=torch.int64, device=self.device)
        else:
            pillars, indices = out
            if (pillars.shape != (B, _lib.NUM_FEATURES, P, N) or pillars.dtype != torch.float32
                    or indices.shape != (B, P, 3) or indices.dtype != torch.int64
                    or not pillars.is_contiguous() or not indices.is_contiguous()):
                raise ValueError("out buffers have the wrong shape/dtype/layout")
        counts = (torch.empty((B, 2), dtype=torch.int32, device=self.device)
                  if return_counts else None)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = _lib.lib().pp_voxelize_dev(
            self._ctx.handle, ctypes.c_void_p(stream), ctypes.c_void_p(points.data_ptr()),
            ncap, n_arr, B, ctypes.byref(self._prm), ctypes.c_void_p(pillars.data_ptr()),
            ctypes.c_void_p(indices.data_ptr()),
            ctypes.c_void_p(counts.data_ptr()) if counts is not None else None)
        _lib.check(rc, "pp_voxelize_dev")
        if self.data_mean is not None:       # dataset.py:102-105
            rc = _lib.lib().pp_subtract_mean_dev(
                self._ctx.handle, ctypes.c_void_p(stream), ctypes.c_void_p(pillars.data_ptr()), B,
                _lib.NUM_FEATURES * P * N, ctypes.c_void_p(self.data_mean.data_ptr()))
            _lib.check(rc, "pp_subtract_mean_dev")
        if return_counts:
            return pillars, indices, counts
        return pillars, indices
