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
    def __init__(self, rank=0, world_size=1, local_rank=0, backend=None):
        self.rank, self.world_size, self.local_rank, self.backend = rank, world_size, local_rank, backend

    @property
    def distributed(self):
        return self.world_size > 1


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract); a
    single process without those variables is world_size 1, no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return ShardContext(rank, world, local, backend)


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


def allreduce_gradients(ctx, parameters, bucket_bytes=32 << 20):
    """Average the gradients over all ranks: what the reference's ``nn.DataParallel`` does
    implicitly (train.py:88-89: one loss over the gathered batch, gradients reduced onto GPU 0).
    The gradients are packed into flat buckets (one all-reduce per ``bucket_bytes``; the whole
    network is 19 MB of f32, i.e. ONE ring all-reduce over xGMI), summed and divided by the
    world size.  Every rank ends with identical gradients, so identical optimizer steps keep the
    replicas in sync without ever broadcasting weights."""
    if not ctx.distributed:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
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
