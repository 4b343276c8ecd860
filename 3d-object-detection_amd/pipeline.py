"""Device-resident sample pipeline: raw points -> pillars -> network (-> loss).

Counterpart of what /root/reference does per step between the lidar file and
the loss: ``PPDataset.__getitem__``'s voxel stage and target stage
(data/dataset.py:88-120) on the HIP kernels, then ``PPModel`` / ``PPLoss``
(train.py:144-145, evaluate.py:228) on PyTorch-ROCm.  Nothing crosses PCIe
except the raw points (0.96 MB per 60k-point sweep) and the boxes.
"""
import torch

from . import boxes
from .loss import PPLoss
from .model import PPModel
from .targets import TargetAssigner
from .voxelizer import PillarVoxelizer, VoxelConfig


class PillarPipeline:
    def __init__(self, vox_cfg: VoxelConfig, anchor_cfg: boxes.AnchorConfig = None,
                 feature_channels=64, num_classes=9, reg_dims=8, device=None, seed=0,
                 pos_thresh=0.6, with_targets=False, data_mean=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.vox_cfg = vox_cfg
        h, w = vox_cfg.canvas_height, vox_cfg.canvas_width
        if anchor_cfg is None:
            anchor_cfg = boxes.AnchorConfig(fm_height=(h + 1) // 2, fm_width=(w + 1) // 2)
        self.anchor_cfg = anchor_cfg
        # data_mean: the optional pillar_means.pkl of train.py:61 / evaluate.py:207
        self.voxelizer = PillarVoxelizer(vox_cfg, device=self.device, data_mean=data_mean)
        torch.manual_seed(seed)  # model/model.py:9
        self.model = PPModel(9, feature_channels, anchor_cfg.per_cell * num_classes,
                             anchor_cfg.per_cell * reg_dims, h, w).to(self.device)
        self.loss = PPLoss()
        self.assigner = None
        if with_targets:
            # anchors evaluated on the fly in the target kernels (no per-anchor arrays)
            self.assigner = TargetAssigner(anchor_cfg, canvas_height=h,
                                           pos_thresh=pos_thresh, num_classes=num_classes,
                                           device=self.device)
        self._bufs = None
        self._fbufs = None
        self._cbuf = None
        self._cidx = None          # the indices of the last fused-scatter call: the canvas's non-zero pixels
        self._canvas_key = None
        #: forward_fused also fuses PPScatter and runs the backbone channels-last
        self.fused_scatter = True

    def _buffers(self, batch):
        cfg = self.vox_cfg
        if self._bufs is None or self._bufs[0].shape[0] != batch:
            self._bufs = (torch.empty((batch, 9, cfg.max_pillars, cfg.max_points_per_pillar),
                                      dtype=torch.float32, device=self.device),
                          torch.empty((batch, cfg.max_pillars, 3), dtype=torch.int64, device=self.device))
        return self._bufs

    def voxelize(self, points, n_points=None):
        if points.dim() == 2:
            points = points.unsqueeze(0)
        return self.voxelizer(points, n_points=n_points, out=self._buffers(points.shape[0]))

    @torch.no_grad()
    def forward(self, points, n_points=None):
        """evaluate.py:216-228: voxel stage + network forward (inference)."""
        pillars, indices = self.voxelize(points, n_points)
        return self.model(pillars, indices)

    @torch.no_grad()
    def forward_pipelined(self, points, n_points=None):
        """The same forward as a software pipeline over consecutive batches (what the reference's
        DataLoader prefetch amounts to, train.py:120-121): the voxelizer's ONE launch per call runs the
        split stage of ``points``, the tile and order stages of the two previous calls' batches and the emit
        stage of the batch before those (``PillarVoxelizer.submit``), and the network runs on that oldest
        batch.  Returns its ``(cls, reg)`` -- or ``None`` for the first ``PillarVoxelizer.LAG`` calls.  ``points=None`` drains."""
        B = self.voxelizer._inflight[-1] if getattr(self.voxelizer, "_inflight", None) else None
        r = self.voxelizer.submit(points, n_points=n_points, out=self._buffers(B) if B else None)
        return None if r is None else self.model(r[0], r[1])

    @torch.no_grad()
    def forward_overlapped(self, points, n_points=None):
        """``forward_pipelined`` with the voxelizer's launch on a SECOND stream, side by side with the network (what
        the reference's DataLoader workers do on CPU cores, train.py:120-121): call i launches ``k_step`` on the side
        stream (it emits an older batch into one of two output buffers) and, on the caller's stream, runs the network
        on the batch the PREVIOUS call's launch emitted into the other buffer.  The two only meet through events that
        were recorded a whole step earlier, so neither queue waits: the voxelizer leaves the step's critical path.
        ``points`` (and ``n_points`` if it is a tensor) may be produced on the caller's stream right before the call
        -- a ``non_blocking`` copy, an ingest kernel -- and dropped right after it: the side stream waits for an event
        recorded on the caller's stream AT ENTRY (between two calls that stream holds only the previous network pass
        and whatever produced ``points``, so the wait also covers "buffer k was read by the previous call's network"
        and costs the overlap nothing), and the tensors are ``record_stream``-ed on the side stream, so the caching
        allocator does not hand their memory to a main-stream allocation while the side-stream launch still reads it.
        Returns ``(cls, reg)`` of the batch handed in ``PillarVoxelizer.LAG + 1`` calls ago, else ``None``;
        ``points=None`` drains."""
        ov = getattr(self, "_ov", None)
        if ov is None:
            ov = self._ov = {"stream": torch.cuda.Stream(device=self.device), "i": 0, "bufs": [None, None],
                             "vox_done": [torch.cuda.Event(), torch.cuda.Event()], "entry": torch.cuda.Event(),
                             "ready": [None, None]}
        main = torch.cuda.current_stream(self.device)
        i, side = ov["i"], ov["stream"]
        k = i % 2
        # everything the caller's stream holds now -- the producer of `points`, the previous call's network pass that
        # read buffer k -- is ordered before the side stream's launch
        ov["entry"].record(main)
        side.wait_event(ov["entry"])
        for t_ in (points, n_points):
            if torch.is_tensor(t_) and t_.is_cuda:
                t_.record_stream(side)
        B = self.voxelizer._inflight[-1] if getattr(self.voxelizer, "_inflight", None) else None
        with torch.cuda.stream(side):
            if B and (ov["bufs"][k] is None or ov["bufs"][k][0].shape[0] != B):
                cfg = self.vox_cfg
                ov["bufs"][k] = (torch.empty((B, 9, cfg.max_pillars, cfg.max_points_per_pillar), dtype=torch.float32,
                                             device=self.device),
                                 torch.empty((B, cfg.max_pillars, 3), dtype=torch.int64, device=self.device))
            r = self.voxelizer.submit(points, n_points=n_points, out=ov["bufs"][k] if B else None)
            ov["vox_done"][k].record(side)
        ov["ready"][k] = r
        ov["i"] = i + 1
        prev = ov["ready"][1 - k]
        ov["ready"][1 - k] = None
        out = None
        if prev is not None:
            main.wait_event(ov["vox_done"][1 - k])
            out = self.model(prev[0], prev[1])
        return out

    @torch.no_grad()
    def forward_fused(self, points, n_points=None):
        """Inference with PPFeatureNet fused into the voxelizer:
        the dense [9,P,N] tensor and the [64,P,N] intermediate never exist.  Needs
        ``model.eval()`` (BatchNorm as an affine map)."""
        if self.model.training:
            raise RuntimeError("forward_fused is inference only: call model.eval() first")
        if self.voxelizer.data_mean is not None:
            # a data mean makes the zero-padded slots non-zero: nothing to skip, use the dense path
            return self.forward(points, n_points)
        pfn_params = self.model.feature_net.fused_table(self.device)   # tracks the weights' versions
        if points.dim() == 2:
            points = points.unsqueeze(0)
        B = points.shape[0]
        P = self.vox_cfg.max_pillars
        if self._fbufs is None or self._fbufs[0].shape[0] != B:
            self._fbufs = (torch.empty((B, 64, P), dtype=torch.float32, device=self.device),
                           torch.empty((B, P, 3), dtype=torch.int64, device=self.device))
        if self.fused_scatter:
            # ... and PPScatter too: the emit kernel writes each pillar's 64 features to its
            # channels-last canvas pixel
            H, W = self.model.scatter.h, self.model.scatter.w
            cbuf = self._canvas(B, H, W)
            # The canvas and ITS index buffer are written by this branch only (the feature path below has its
            # own indices): after the first call only the previous call's pixels are non-zero, and only those
            # are cleared again.  The key is dropped before the call, so an error cannot leave a stale one.
            if self._cidx is None or self._cidx.shape[0] != B:
                self._cidx = torch.empty((B, P, 3), dtype=torch.int64, device=self.device)
                self._canvas_key = None
            key = (cbuf.data_ptr(), self._cidx.data_ptr(), B)
            reuse = self._canvas_key == key
            self._canvas_key = None
            canvas, _ = self.voxelizer.pfn_canvas(points, pfn_params, (H, W), n_points=n_points,
                                                  out=(cbuf, self._cidx), reuse=reuse)
            self._canvas_key = key
            return self.model.forward_canvas(canvas)
        feats, indices = self.voxelizer.pfn(points, pfn_params, n_points=n_points, out=self._fbufs)
        return self.model.forward_features(feats, indices)

    @torch.no_grad()
    def forward_fused_pipelined(self, points, n_points=None):
        """``forward_fused`` as a software pipeline over consecutive batches: ONE voxelizer launch per call
        (``PillarVoxelizer.submit_pfn_canvas``: the split stage of ``points``, the tile and order stages of the two
        previous batches, and the emit stage of the batch before those AS the fused feature net writing straight
        into the channels-last canvas), and the network runs on that oldest batch's canvas.  Returns its
        ``(cls, reg)`` -- ``None`` for the first ``PillarVoxelizer.LAG`` calls; ``points=None`` drains."""
        if self.model.training:
            raise RuntimeError("forward_fused_pipelined is inference only: call model.eval() first")
        if self.voxelizer.data_mean is not None:
            # as forward_fused: a data mean makes the zero-padded slots non-zero, nothing to skip -- the dense
            # pipelined path (same pipeline, same lag: the batches in flight are shared between the two forms)
            return self.forward_pipelined(points, n_points)
        pfn_params = self.model.feature_net.fused_table(self.device)
        H, W = self.model.scatter.h, self.model.scatter.w
        r = self.voxelizer.submit_pfn_canvas(points, pfn_params, (H, W), n_points=n_points, channels_last=True)
        return None if r is None else self.model.forward_canvas(r[0])

    def _canvas(self, B, H, W):
        if self._cbuf is None or self._cbuf.shape != (B, 64, H, W):
            self._cbuf = torch.empty((B, 64, H, W), dtype=torch.float32, device=self.device,
                                     memory_format=torch.channels_last)
        return self._cbuf

    def upload_ground_truth(self, g):
        """Host box arrays (centers / wlh / yaw / classes, canvas space) -> the device tuple
        ``TargetAssigner.assign_device`` consumes (image-space corners are derived here,
        utils/box_utils.py:19-32)."""
        return self.assigner._gt_to_device(g["centers"], g["wlh"], g["yaw"], g["classes"])

    def upload_ground_truth_batch(self, gts):
        """The boxes of ALL samples of a step in one device buffer (one host-to-device copy):
        ``(g_counts, packed)``, what ``TargetAssigner.assign_batch_device`` and
        ``train_forward_backward`` consume."""
        return self.assigner.upload_batch(gts)

    def train_forward_backward(self, points, gts, n_points=None, shard_ctx=None):
        """train.py:139-147 without the optimizer: voxel stage, target stage, forward,
        loss, backward.  ``gts`` is a list (one per sweep) of dicts with
        centers / wlh / yaw / classes in canvas space.  With a multi-rank ``shard_ctx`` the
        back-propagated loss is ``shard.global_batch_loss`` (this rank's share of the ONE loss
        the reference computes over the gathered batch); the returned scalars stay the local
        PPLoss values."""
        pillars, indices = self.voxelize(points, n_points)
        if isinstance(gts, tuple) and len(gts) == 2 and torch.is_tensor(gts[1]):
            cls_t, reg_t = self.assigner.assign_batch_device(*gts)
        elif all(isinstance(g, dict) for g in gts):
            cls_t, reg_t = self.assigner.assign_batch(gts)
        else:
            # per-sample device tuples of upload_ground_truth(): one launch per sample
            targets = [self.assigner.assign_device(*g) if isinstance(g, tuple)
                       else self.assigner.assign(g["centers"], g["wlh"], g["yaw"], g["classes"]) for g in gts]
            cls_t = torch.stack([t[0] for t in targets])
            reg_t = torch.stack([t[1] for t in targets])
        cls, reg = self.model(pillars, indices)
        p, cls_loss, reg_loss, ort_loss, total = self.loss(cls, reg, cls_t, reg_t)
        if shard_ctx is not None and shard_ctx.distributed:
            from . import shard
            n_pos = (reg_t[..., 0] == 1).sum()
            shard.global_batch_loss(shard_ctx, self.loss, cls_loss, reg_loss, ort_loss, n_pos).backward()
        else:
            total.backward()
        return cls_loss.detach(), reg_loss.detach(), ort_loss.detach(), total.detach()
