"""RCCL itself, on the one GPU a box of this pool has: a communicator of ONE rank (two ranks on one device are refused by
the library: "duplicate GPU").  What it shows that the gloo rehearsals cannot: the `nccl` backend of this image
initialises under the environment shard.init_from_env leaves (HSA_ENABLE_IPC_MODE_LEGACY=0), binds to the device, and
carries every collective the N > 1 run issues -- the loss-scalar and positive-count all-reduces, the bucketed gradient
all-reduce of the whole network (19 MB of f32: one call), the max-over-ranks timing reduction, bench.py's per-rank
all-gather and the barrier -- through ncclAllReduce / ncclAllGather on device buffers with the right results.  The
eight-GPU run over xGMI is the driver's.  (/root/reference train.py:88-89,120-121,144-147.)"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
import pp_amd
from pp_amd import shard
from pp_amd.loss import PPLoss
from pp_amd.model import PPModel

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"


class OneRank(shard.ShardContext):          # world size 1, but every collective is issued (as at N > 1)
    distributed = property(lambda self: True)


ctx = OneRank(0, 1, 0, "nccl")
# the four loss scalars, sweep-weighted (shard.reduce_loss_scalars): one all-reduce of 5 floats
out = shard.reduce_loss_scalars(ctx, 1.5, 2.5, 3.5, 7.5, 4, device=dev)
assert out.is_cuda and torch.allclose(out.cpu(), torch.tensor([1.5, 2.5, 3.5, 7.5]))
# positive counts before the backward (shard.global_batch_loss)
loss = PPLoss()
c, r, o = (torch.tensor(v, device=dev, requires_grad=True) for v in (0.4, 0.2, 0.1))
tot = shard.global_batch_loss(ctx, loss, c, r, o, 31)
want = loss.b_cls * 0.4 + loss.b_reg * 0.2 + loss.b_ort * 0.1      # every positive is this rank's: scale 1
assert abs(float(tot.detach()) - want) < 1e-5 * want
# the whole network's gradients in flat buckets: averaging over one rank leaves them as they are
torch.manual_seed(0)
net = PPModel(9, 64, 2 * 9, 2 * 8, canvas_height=500, canvas_width=500).to(dev)
for p in net.parameters():
    p.grad = torch.randn_like(p)
before = [p.grad.clone() for p in net.parameters()]
nbytes = sum(g.numel() * 4 for g in before)
calls = shard.allreduce_gradients(ctx, net.parameters())
assert calls == 1 and nbytes < (32 << 20)          # the whole network: ONE ring all-reduce
assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
# timing reduction, barrier, and the [world, k] gather bench.py's per-rank records use
assert shard.max_over_ranks(ctx, 0.125, device=dev) == 0.125
shard.barrier(ctx)
mine = torch.tensor([[34.2, 11.3, 0.65]], device=dev)
got = [torch.empty_like(mine)]
dist.all_gather(got, mine)
assert torch.equal(got[0], mine)
torch.cuda.synchronize()
print("rccl one rank ok", torch.cuda.nccl.version(), "gradient bytes", nbytes, "all-reduce calls", calls, flush=True)
shard.shutdown(ctx)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_rccl_communicator_of_one_rank_carries_every_collective(gpu, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), TMPDIR=str(tmp_path))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
