"""GPU tests of the PyTorch-ROCm network on HIP-voxelized input and of bench.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_on_gpu_matches_golden(gpu):
    import torch
    import pp_amd.model as M
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_golden.npz"))
    canvas, c, p, n, a_per = [int(v) for v in g["dims"]]
    net = M.PPModel(9, c, a_per * 9, a_per * 8, canvas, canvas)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    net = net.to(gpu).eval()
    with torch.no_grad():
        cl, rg = net(torch.from_numpy(g["pillars"]).to(gpu), torch.from_numpy(g["inds"]).to(gpu))
    assert np.abs(cl.cpu().numpy() - g["cls_eval"]).max() < 1e-4      # f32 conv, different summation order
    assert np.abs(rg.cpu().numpy() - g["reg_eval"]).max() < 1e-4


def test_pipeline_forward_equals_model_on_oracle_pillars(gpu, oracle):
    """End to end: HIP voxelizer + network == network on the oracle's voxel stage."""
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    from util import oracle_stage
    cfg = VoxelConfig.square(16.0, 0.2, 4000, 32)
    pipe = PillarPipeline(cfg, feature_channels=16, device=gpu, seed=0)
    pipe.model.eval()
    pts = synth.lidar_like(15000, 16.0, 3)
    cl, rg = pipe.forward(torch.from_numpy(pts).to(gpu))
    ref_p, ref_i, _ = oracle_stage(oracle, pts, 4000, 32, 16.0, 0.2)
    with torch.no_grad():
        cl2, rg2 = pipe.model(torch.from_numpy(ref_p)[None].to(gpu), torch.from_numpy(ref_i)[None].to(gpu))
    assert cl.shape == (1, 18, 80, 80) and rg.shape == (1, 16, 80, 80)
    assert torch.equal(cl, cl2) and torch.equal(rg, rg2)


def test_train_step_runs_and_is_finite(gpu):
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    cfg = VoxelConfig.square(16.0, 0.2, 3000, 16)
    pipe = PillarPipeline(cfg, feature_channels=8, device=gpu, seed=0, with_targets=True)
    pipe.model.train()
    pts = torch.from_numpy(np.stack([synth.lidar_like(12000, 16.0, s) for s in (0, 1)])).to(gpu)
    gts = [synth.gt_boxes(6, 160, s, margin=25.0) for s in (0, 1)]
    cl, rl, ol, tot = pipe.train_forward_backward(pts, gts)
    assert all(torch.isfinite(v) for v in (cl, rl, ol, tot))
    grads = [p.grad for p in pipe.model.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)


def test_bench_contract_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 0 and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
