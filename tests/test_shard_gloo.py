"""world_size-2 test of the multi-process path on CPU (gloo): sweep sharding,
the loss-scalar all-reduce and the max-over-ranks timing reduction."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pp_amd import shard
    ctx = shard.init_from_env("gloo")
    mine = shard.sweeps_for_rank(5, ctx.rank, ctx.world_size)
    # per-rank "losses": functions of the owned sweep ids
    vals = [float(sum(mine)) + k for k in range(4)]
    red = shard.reduce_loss_scalars(ctx, *vals, n_local=len(mine))
    tmax = shard.max_over_ranks(ctx, 1.0 + rank)
    # gradient averaging (the DataParallel semantics of train.py:88-89): two buckets
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    x = torch.full((4, 6), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    calls = shard.allreduce_gradients(ctx, net.parameters(), bucket_bytes=100)
    shard.barrier(ctx)
    q.put((rank, mine, red.tolist(), tmax, [g.tolist() for g in local],
           [p.grad.tolist() for p in net.parameters()], calls))
    shard.shutdown(ctx)


def test_two_rank_shard_and_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]
    # sweep-weighted mean: (3*(3+k) + 2*(7+k)) / 5
    exp = [(3 * (3 + k) + 2 * (7 + k)) / 5 for k in range(4)]
    for r in res:
        assert torch.allclose(torch.tensor(r[2]), torch.tensor(exp), atol=1e-6)
        assert r[3] == 2.0
    # every rank holds the mean of the two local gradients, in every bucket
    assert res[0][6] >= 2 and res[0][6] == res[1][6]
    for k in range(len(res[0][4])):
        mean = (torch.tensor(res[0][4][k]) + torch.tensor(res[1][4][k])) / 2
        for r in res:
            assert torch.allclose(torch.tensor(r[5][k]), mean, atol=1e-6)


def test_single_process_is_world_one(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    from pp_amd import shard
    ctx = shard.init_from_env("gloo")
    assert (ctx.rank, ctx.world_size, ctx.distributed) == (0, 1, False)
    red = shard.reduce_loss_scalars(ctx, 1.0, 2.0, 3.0, 4.0, n_local=2)
    assert red.tolist() == [1.0, 2.0, 3.0, 4.0] and shard.max_over_ranks(ctx, 0.5) == 0.5


def _dp_inputs(zero_pos_rank=None):
    """Two 'sweeps' per rank of a toy head: cls [B,2*9,H,W], reg [B,2*8,H,W] from one shared
    weight; targets with a different number of positives per rank."""
    g = torch.Generator().manual_seed(7)
    B, H, W, Ac = 2, 3, 4, 2
    x = [torch.randn(B, 5, H, W, generator=g) for _ in range(2)]
    cls_t = [(torch.rand(B, H * W * Ac, 9, generator=g) > 0.8).float() for _ in range(2)]
    reg_t = []
    for r in range(2):
        t = torch.randn(B, H * W * Ac, 9, generator=g)
        t[..., 0] = (torch.rand(B, H * W * Ac, generator=g) > (0.5 if r == 0 else 0.9)).float()
        t[..., 8] = (t[..., 8] > 0).float()
        if zero_pos_rank == r:
            t[..., 0] = 0
        reg_t.append(t)
    return x, cls_t, reg_t


def _toy_head():
    torch.manual_seed(3)
    return torch.nn.Conv2d(5, 2 * 9 + 2 * 8, 1)


def _dp_worker(rank, world, port, q, zero_pos_rank):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pp_amd import shard
    from pp_amd.loss import PPLoss
    ctx = shard.init_from_env("gloo")
    x, cls_t, reg_t = _dp_inputs(zero_pos_rank)
    head, loss = _toy_head(), PPLoss(b_ort=0.5)
    y = head(x[rank])
    _, c, r, o, _ = loss(y[:, :18], y[:, 18:], cls_t[rank], reg_t[rank])
    n_pos = (reg_t[rank][..., 0] == 1).sum()
    shard.global_batch_loss(ctx, loss, c, r, o, n_pos).backward()
    shard.allreduce_gradients(ctx, head.parameters())
    q.put((rank, [p.grad.tolist() for p in head.parameters()]))
    shard.barrier(ctx)
    shard.shutdown(ctx)


def _run_dp(zero_pos_rank):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, zero_pos_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the reference: ONE loss over the gathered batch (nn.DataParallel, train.py:88-89,144-147)
    from pp_amd.loss import PPLoss
    x, cls_t, reg_t = _dp_inputs(zero_pos_rank)
    head, loss = _toy_head(), PPLoss(b_ort=0.5)
    y = head(torch.cat(x))
    loss(y[:, :18], y[:, 18:], torch.cat(cls_t), torch.cat(reg_t))[4].backward()
    for rank, grads in res:
        for g, p in zip(grads, head.parameters()):
            assert torch.isfinite(torch.tensor(g)).all()
            assert torch.allclose(torch.tensor(g), p.grad, rtol=1e-5, atol=1e-6), rank


def test_sharded_gradients_equal_the_gathered_batch_loss():
    """Ranks with different positive counts: weighting the positives' means by
    n_pos_local * world / n_pos_global makes the averaged gradients those of the reference's
    single loss over the gathered batch (plain averaging mis-weights them)."""
    _run_dp(None)


def test_rank_without_positives_contributes_zero_not_nan():
    _run_dp(1)


def _bench_ranks_worker(rank, world, port, q):
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pp_amd import shard
    spec = importlib.util.spec_from_file_location("pp_bench_ranks", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = b
    spec.loader.exec_module(b)
    ctx = shard.init_from_env("gloo")
    rows = b.per_rank(ctx, [10.0 + rank, 100.0 * (rank + 1)])
    mm = b.ranks_min_max(ctx, 30.0 + 10.0 * rank, 8.0e12 * 30e-6 * 0.5)     # rank 0: 0.5 of the peak, rank 1: 0.375
    q.put((rank, rows, mm))
    shard.barrier(ctx)
    shard.shutdown(ctx)


def test_bench_per_rank_records_over_two_ranks():
    """bench.py's per-rank roofline spread (roofline.ranks_min_max, stress_c5.per_rank, configs3 records): every rank
    ends up with every rank's kernel time, min / max and the fractions, through ONE all-reduce after the loop."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_ranks_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, rows, mm in res:
        assert rows == [[10.0, 100.0], [11.0, 200.0]]
        assert mm["k_step_us"] == [30.0, 40.0] and mm["per_rank_k_step_us"] == [30.0, 40.0]
        assert abs(mm["frac"][0] - 0.375) < 1e-12 and abs(mm["frac"][1] - 0.5) < 1e-12


def _eight_worker(rank, world, port, q):
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from pp_amd import shard
    spec = importlib.util.spec_from_file_location("pp_bench_ranks8", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = b
    spec.loader.exec_module(b)
    ctx = shard.init_from_env("gloo")
    mine = shard.sweeps_for_rank(8, ctx.rank, ctx.world_size)               # configs[3]: batch = 8, one sweep per GPU
    red = shard.reduce_loss_scalars(ctx, float(rank), 2.0 * rank, 0.0, 3.0 * rank, n_local=len(mine))
    tmax = shard.max_over_ranks(ctx, 1.0 + 0.25 * rank)
    mm = b.ranks_min_max(ctx, 13.0 + rank, 44_448_000)
    torch.manual_seed(0)
    net = torch.nn.Linear(4, 3)
    net(torch.full((2, 4), float(rank))).sum().backward()
    shard.allreduce_gradients(ctx, net.parameters(), bucket_bytes=32)
    shard.barrier(ctx)
    q.put((rank, mine, red.tolist(), tmax, mm, net.weight.grad.tolist()))
    shard.shutdown(ctx)


def test_eight_ranks_on_cpu_rehearse_configs3():
    """BASELINE configs[3] is 8 ranks, one sweep each (/root/reference train.py:88-89,120-121 as one process per
    GPU).  Eight GPU processes cannot share this pool's one-GPU box (its process guard allows six), so the eight-rank
    code path is rehearsed here over gloo on the CPU: one sweep id per rank, the loss-scalar all-reduce weighted by
    sweeps, the max-over-ranks clock, bench.py's per-rank roofline record of length 8, bucketed gradient averaging."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[k] for k in range(8)]                    # rank k owns sweep k
    mean_rank = sum(range(8)) / 8.0
    for rank, _, red, tmax, mm, grad in res:
        assert torch.allclose(torch.tensor(red), torch.tensor([mean_rank, 2 * mean_rank, 0.0, 3 * mean_rank]), atol=1e-6)
        assert tmax == 1.0 + 0.25 * 7
        assert mm["per_rank_k_step_us"] == [13.0 + k for k in range(8)] and mm["k_step_us"] == [13.0, 20.0]
        assert len(mm["frac"]) == 2 and mm["frac"][0] < mm["frac"][1]
        assert torch.allclose(torch.tensor(grad), torch.full((3, 4), 2.0 * mean_rank), atol=1e-6)   # mean over ranks of 2 * rank
