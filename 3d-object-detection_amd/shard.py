"""One process per GPU: sweep sharding and the loss-scalar exchange.

The reference scales with single-process ``nn.DataParallel`` (/root/reference
train.py:88-89), which moves every sample's dense input through GPU 0.  Here
each rank owns whole sweeps (they are independent units: data/dataset.py:37-122
is per index), voxelizes and forwards them on its own device, and the only
exchange is one all-reduce of the four loss scalars (16 bytes) -- RCCL over xGMI
on GPUs (``backend="nccl"``), gloo in the CPU tests.  No data-path collective.
"""
import os

import torch
import torch.distributed as dist


class ShardContext:
    def __init__(self, rank=0, world_size=1, local_rank=0, backend=None, force_group=False):
        self.rank, self.world_size, self.local_rank, self.backend = rank, world_size, local_rank, backend
        self.force_group = force_group

    @property
    def distributed(self):
        """True when the collectives are issued: more than one rank -- or a process group of ONE rank that was asked
        for explicitly (PP_FORCE_PROCESS_GROUP=1: the N > 1 code path, every all-reduce and barrier of it, on a box
        with one GPU; the rehearsal of RCCL itself that two ranks on one device cannot give)."""
        return self.world_size > 1 or self.force_group


def private_miopen_cache(local_rank, root=None):
    """Give this process its own MIOpen user database / kernel cache directory.  N ranks that
    start together on a fresh box would otherwise run their first convolutions' find/compile
    step against ONE shared user find-db and write it concurrently.  Must run before the first
    convolution of the process; explicit MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR win."""
    root = root or os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pp_miopen_{os.getuid()}")
    d = os.path.join(root, f"rank{int(local_rank)}")
    for var, sub in (("MIOPEN_USER_DB_PATH", "udb"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if var not in os.environ:
            path = os.path.join(d, sub)
            os.makedirs(path, exist_ok=True)
            os.environ[var] = path
    return d


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract); a
    single process without those variables is world_size 1, no process group
    (unless PP_FORCE_PROCESS_GROUP=1 asks for a group of one rank: ShardContext.distributed)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = world == 1 and os.environ.get("PP_FORCE_PROCESS_GROUP") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        # The pool's operating notes: the host driver supports dmabuf IPC only, and RCCL needs this variable where the
        # environment does not already carry it -- set if absent, never overridden.  HIP / HSA read their environment
        # when they initialise, so this is done BEFORE anything below touches the device (torch.cuda.is_available()
        # does); it only helps if nothing in the process has initialised HIP yet (bench.py sets it at import too).
        if backend in (None, "nccl"):
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            # bind the communicator to this rank's GPU up front (no lazy device guess at the
            # first collective, no barrier-on-wrong-device warning)
            kw["device_id"] = torch.device("cuda", local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force:
            os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return ShardContext(rank, world, local, backend, force_group=force and dist.is_initialized())


def sweeps_for_rank(num_sweeps, rank, world_size):
    """Global sweep ids owned by ``rank``: contiguous blocks, remainder to the
    first ranks.  With num_sweeps == world_size this is one sweep per GPU
    (BASELINE config 4)."""
    base, rem = divmod(num_sweeps, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def reduce_loss_scalars(ctx, cls_loss, reg_loss, ort_loss, total, n_local, device=None):
    """Sweep-weighted mean of [cls, reg, ort, total] over all ranks: one
    all-reduce(SUM) of a 5-float tensor (4 scalars + sweep count)."""
    vals = torch.stack([torch.as_tensor(v, dtype=torch.float32, device=device).reshape(())
                        for v in (cls_loss, reg_loss, ort_loss, total)])
    buf = torch.cat([vals.detach() * float(n_local),
                     torch.tensor([float(n_local)], dtype=torch.float32, device=vals.device)])
    if ctx.distributed:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[:4] / buf[4].clamp_min(1.0)


def global_batch_loss(ctx, loss, cls_loss, reg_loss, ort_loss, n_pos_local):
    """The loss to back-propagate on this rank so that AVERAGING the ranks' gradients
    (allreduce_gradients) gives the gradient of the reference's one loss over the gathered
    batch (nn.DataParallel, train.py:88-89,144-147).

    PPLoss's classification term is a mean over all anchors: equal sweeps per rank make the
    average of the ranks' means the global mean.  The regression and orientation terms are
    means over the POSITIVE anchors (model/loss.py:53-62): rank r's term must be weighted by
    n_pos_r * world / n_pos_global -- one all-reduce of the positive counts, before the
    backward.  A rank without positives contributes zero (its local mean over an empty
    selection is NaN in the reference too; there only if the WHOLE batch has none -- with no
    positive anywhere this returns a finite loss where the reference returns NaN).
    ``loss`` is the PPLoss module (its b_cls / b_reg / b_ort)."""
    n_local = torch.as_tensor(n_pos_local, dtype=torch.float32, device=cls_loss.device).reshape(1)
    n_global = n_local.clone()
    if ctx.distributed:
        dist.all_reduce(n_global, op=dist.ReduceOp.SUM)
    scale = (n_local * float(ctx.world_size) / n_global.clamp_min(1.0)).reshape(())
    have = n_local.reshape(()) > 0
    zero = torch.zeros((), dtype=cls_loss.dtype, device=cls_loss.device)
    reg = torch.where(have, reg_loss, zero) * scale
    ort = torch.where(have, ort_loss, zero) * scale
    return loss.b_cls * cls_loss + loss.b_reg * reg + loss.b_ort * ort


def allreduce_gradients(ctx, parameters, bucket_bytes=32 << 20):
    """Average the gradients over all ranks: with ``global_batch_loss`` as each rank's loss this
    is the reference's ``nn.DataParallel`` step (train.py:88-89: one loss over the gathered
    batch, gradients reduced onto GPU 0).  The gradients are packed into flat buckets (one
    all-reduce per ``bucket_bytes``; the whole network is 19 MB of f32, i.e. ONE ring
    all-reduce over xGMI), summed and divided by the world size.  EVERY parameter takes part
    (a missing gradient counts as zeros), so all ranks issue the same collectives whatever
    their local graph touched.  Every rank ends with identical gradients, so identical
    optimizer steps keep the replicas in sync without ever broadcasting weights."""
    if not ctx.distributed:
        return 0
    grads = []
    for p in parameters:
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    n_calls, i = 0, 0
    while i < len(grads):
        bucket, size = [], 0
        while i < len(grads) and (not bucket or size + grads[i].numel() * grads[i].element_size() <= bucket_bytes):
            if bucket and grads[i].dtype != bucket[0].dtype:
                break
            bucket.append(grads[i])
            size += grads[i].numel() * grads[i].element_size()
            i += 1
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ctx.world_size)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_calls += 1
    return n_calls


def max_over_ranks(ctx, seconds, device=None):
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if ctx.distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(ctx):
    if ctx.distributed:
        dist.barrier()


def shutdown(ctx):
    if ctx.distributed and dist.is_initialized():
        dist.destroy_process_group()
