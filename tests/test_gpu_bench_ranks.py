"""bench.py's multi-rank path on one GPU: two ranks over gloo sharing device 0 (RCCL needs
one GPU per rank; the driver runs that on the 8-GPU node).  Exercises what the N > 1 run does
that N = 1 does not: the process group, per-rank MIOpen directories, sweep sharding, barrier +
max-over-ranks timing, and -- in the train_c3 leg / --mode train -- the positive-count, gradient
and loss-scalar all-reduces of BASELINE configs[2]/[3] (/root/reference train.py:88-89,120-121)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fwd", "train"])
def test_two_ranks_gloo_shared_device(gpu, mode, tmp_path):
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
            "--train-steps", "2", "--backend", "gloo", "--mode", mode, "--no-cpu-baseline"]
    if mode == "train":
        args.append("--no-stress")
    if mode == "fwd":
        # the way the driver starts it: plain `python bench.py --gpus N`, no launcher, no WORLD_SIZE --
        # bench.py starts the N rank processes itself (before any GPU call in the parent)
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    env = dict(os.environ, TMPDIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 2
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["global_batch"] == 2 * out["config"]["sweeps_per_gpu_per_step"]
    assert out["value"] > 0 and abs(out["value"] - 3 * out["config"]["global_batch"] /
                                    (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert out["collectives"]["world_size"] == 2 and out["collectives"]["backend"] == "gloo"
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1 and 0 < rf["pipeline_frac"] <= rf["frac"]
    if mode == "fwd":      # software-pipelined voxelizer: one kernel; the three-launch path beside it
        assert rf["kernel"].startswith("pp::k_step")
        assert set(rf["three_launch"]["kernels_us"]) == {"k_split", "k_tile", "k_emit"}
    else:
        assert set(rf["pipeline"]["kernels_us"]) == {"k_split", "k_tile", "k_emit"}
    if mode == "fwd":
        tr = out["train_c3"]
        assert tr["collectives"]["world_size"] == 2 and tr["collectives"]["backend"] == "gloo"
        assert tr["allreduce_ms"] > 0 and tr["ms_per_step"] > 0
        assert tr["targets"]["bytes_per_launch"] == 112 * 125000 * 4 and tr["targets"]["us_per_call"] > 0
        # BASELINE configs[3] as named: one sweep per GPU and step, forward and training step, with its collectives
        c3 = tr["configs3_one_sweep_per_gpu"]
        assert c3["global_batch"] == 2 and c3["sweeps_per_gpu_per_step"] == 1
        assert c3["collectives"]["world_size"] == 2 and c3["collectives"]["backend"] == "gloo"
        for leg in ("forward", "train"):
            assert c3[leg]["value"] > 0 and abs(c3[leg]["value"] - 2 * c3[leg]["steps"] /
                                               (c3[leg]["ms_per_step"] * c3[leg]["steps"] * 1e-3)) < 1e-6 * c3[leg]["value"]
        assert c3["allreduce_ms"] > 0
        # ... with the roofline records of the two kernels every rank runs at one sweep per step, min / max over ranks
        r3 = c3["roofline"]
        assert r3["kernel"].startswith("pp::k_step") and r3["bytes_per_launch"] == 44_448_000
        mm = r3["ranks_min_max"]
        assert len(mm["per_rank_avg_launch_us"]) == 2 and mm["avg_launch_us"][0] <= mm["avg_launch_us"][1]
        assert 0 < mm["frac"][0] <= mm["frac"][1] <= 1
        assert abs(mm["per_rank_avg_launch_us"][0] - r3["avg_launch_us"]) < 1e-6     # rank 0 prints its own
        t3 = c3["targets"]
        assert t3["bytes_per_launch"] == 112 * 125000 and t3["moved_bytes"] == 72 * 125000
        assert len(t3["ranks_min_max"]["per_rank_us_per_call"]) == 2 and t3["us_per_call"] > 0
        # the headline's own spread over the ranks
        hm = rf["ranks_min_max"]
        assert len(hm["per_rank_avg_launch_us"]) == 2 and abs(hm["per_rank_avg_launch_us"][0] - rf["avg_launch_us"]) < 1e-6
        assert hm["frac"][0] <= rf["frac"] + 1e-9 <= hm["frac"][1] + 2e-9
        # BASELINE configs[4] ("8xMI355X stress") at N > 1: every rank its own 200k-point sweeps, whole-job rate,
        # per-rank kernel fractions; and the 1000x1000 canvas end to end
        s5 = out["stress_c5"]
        assert s5["n_gpus"] == 2 and "each of 2 GPUs" in s5["workload"]
        pr = s5["per_rank"]
        assert len(pr["per_rank_k_step_us"]) == 2 and pr["k_step_us"][0] <= pr["k_step_us"][1]
        assert 0 < pr["kernel_frac"][0] <= pr["kernel_frac"][1] <= 1
        assert abs(s5["pipelined"]["sweeps_per_s"] - 2 * out["config"]["sweeps_per_gpu_per_step"] /
                   (s5["pipelined"]["us_per_step"] * 1e-6)) < 1e-6 * s5["pipelined"]["sweeps_per_s"]
        assert s5["one_sweep_per_launch"]["n_gpus"] == 2 and len(
            s5["one_sweep_per_launch"]["per_rank"]["per_rank_k_step_us"]) == 2
        e5 = s5["end_to_end"]
        assert e5["value"] > 0 and e5["up3_output_padding"] == 3 and 0 < e5["k_step_frac"] <= 1
        assert out["pipelined"] is True and out["latency_calls"] == 3 and out["three_launch_value"] > 0
        assert out["overlapped"]["value"] > 0 and out["overlapped"]["latency_calls"] == 4
        assert "cpu_baseline" not in out and "dropin_host" not in out and "reference_default" not in out   # N = 1 legs only
    # every rank used its own MIOpen directories under TMPDIR
    roots = [d for d in os.listdir(tmp_path) if d.startswith("pp_miopen_")]
    assert roots and sorted(os.listdir(os.path.join(tmp_path, roots[0]))) == ["rank0", "rank1"]


@pytest.mark.gpu
def test_four_ranks_gloo_shared_device_forward(gpu, tmp_path):
    """The widest multi-rank rehearsal this pool's one-GPU box allows: its process guard admits six processes on the
    card, this test process is one of them, so FOUR ranks share device 0 over gloo (the eight-rank arithmetic of
    BASELINE configs[3] is rehearsed on the CPU: tests/test_shard_gloo.py::test_eight_ranks_on_cpu_rehearse_configs3;
    the eight-GPU curve is the driver's).  `python bench.py --gpus 4`, no launcher: the contract line for N = 4, the
    process group's size, four private MIOpen directories, per-rank kernel times of length 4, inside the driver's
    per-N time budget.  (/root/reference train.py:88-89,120-121.)"""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "2",
           "--backend", "gloo", "--no-cpu-baseline", "--no-stress", "--no-train-leg"]
    env = dict(os.environ, TMPDIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 4 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["collectives"]["world_size"] == 4 and out["collectives"]["backend"] == "gloo"
    assert out["config"]["global_batch"] == 4 * out["config"]["sweeps_per_gpu_per_step"]
    assert out["value"] > 0 and abs(out["value"] - 3 * out["config"]["global_batch"] /
                                    (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    mm = out["roofline"]["ranks_min_max"]
    assert len(mm["per_rank_avg_launch_us"]) == 4 and mm["avg_launch_us"][0] <= mm["avg_launch_us"][1]
    assert "train_c3" not in out and "stress_c5" not in out and "cpu_baseline" not in out
    assert out["cpu_baseline_from"].startswith("the N = 1 line")
    roots = [d for d in os.listdir(tmp_path) if d.startswith("pp_miopen_")]
    assert roots and sorted(os.listdir(os.path.join(tmp_path, roots[0]))) == ["rank0", "rank1", "rank2", "rank3"]
    assert wall < 600, f"{wall:.0f} s for the N = 4 line"


@pytest.mark.gpu
def test_one_rank_over_rccl_runs_the_multi_rank_code_path(gpu, tmp_path):
    """RCCL under bench.py: `PP_FORCE_PROCESS_GROUP=1 python bench.py --gpus 1 --backend nccl` makes a process group of ONE
    rank on the `nccl` backend and takes the N > 1 code path through it -- barriers around the timed loops, the
    max-over-ranks reduction, the per-rank all-reduces behind `ranks_min_max`, and the train leg's positive-count,
    gradient and loss-scalar all-reduces -- on the one GPU of a box (two ranks on one device are refused by RCCL; the
    gloo rehearsals above cover world sizes 2 and 4).  The line must be the N = 1 line with a `collectives` record that
    names RCCL.  (/root/reference train.py:88-89,120-121.)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--train-steps", "2",
           "--backend", "nccl", "--no-cpu-baseline", "--no-stress", "--no-dropin", "--no-next-rows", "--no-live-traffic",
           "--no-fused"]
    env = dict(os.environ, TMPDIR=str(tmp_path), PP_FORCE_PROCESS_GROUP="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["collectives"]["backend"] == "nccl" and out["collectives"]["world_size"] == 1
    tr = out["train_c3"]
    assert tr["collectives"]["backend"] == "nccl" and tr["collectives"]["world_size"] == 1
    assert tr["allreduce_ms"] > 0                         # the gradient buckets went through ncclAllReduce
    assert len(out["roofline"]["ranks_min_max"]["per_rank_avg_launch_us"]) == 1
