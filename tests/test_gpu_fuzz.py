"""Randomised GPU parity sweep: many small (grid, P, N, density, order) draws,
dense output AND fused-feature-net output against the oracle / PyTorch.  Aims at
the rarely taken paths: LDS-pool overflow with medium buckets (zero pass + late
passes), ballot-rescan pillars, N not a multiple of 4, plane strides that are not
a multiple of 8 groups (group-granular deferral), rows longer than the line mask,
P smaller / larger than the number of occupied cells, both pillar orders."""
import numpy as np
import pytest

from util import grid_args

pytestmark = pytest.mark.gpu


def _cloud(rng, n, half, hot):
    """Uniform background + `hot` dense blobs (buckets of tens to hundreds of points)."""
    pts = np.empty((n, 4), np.float32)
    pts[:, :2] = rng.uniform(-half, half, (n, 2))
    pts[:, 2] = rng.uniform(-2, 2, n)
    pts[:, 3] = rng.uniform(0.01, 255, n)
    k = 0
    for _ in range(hot):
        m = int(rng.integers(20, 400))
        c = rng.uniform(-half * 0.8, half * 0.8, 2)
        sl = slice(k, min(n, k + m))
        pts[sl, :2] = c + rng.normal(0, rng.uniform(0.02, 0.3), (sl.stop - sl.start, 2))
        k = sl.stop
        if k >= n:
            break
    rng.shuffle(pts)
    return pts


CASES = list(range(24))


@pytest.mark.parametrize("case", CASES)
def test_random_configuration(gpu, oracle, case):
    import torch
    import pp_amd.model as M
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    rng = np.random.default_rng(1000 + case)
    half = float(rng.choice([4.0, 6.4, 10.0]))
    step = float(rng.choice([0.2, 0.4, 0.5]))
    N = int(rng.choice([4, 6, 8, 12, 20, 32, 100, 260]))
    n = int(rng.integers(500, 12000))
    hot = int(rng.integers(0, 12))
    order = int(rng.integers(0, 2))
    pts = _cloud(rng, n, half, hot)
    cc = oracle.cell_counts(pts.astype(np.float64), *grid_args(half, step))
    P = int(max(1, len(cc) * rng.choice([0.3, 0.9, 1.0, 1.5]) + rng.integers(0, 5)))
    cfg = VoxelConfig.square(half, step, P, N, order=order)
    vox = PillarVoxelizer(cfg, device=gpu)
    t = torch.from_numpy(pts).to(gpu)
    pil, idx, cnt = vox(t, return_counts=True)
    torch.cuda.synchronize()
    ref_p, ref_i, m = oracle.dataset_voxel_stage(pts.astype(np.float64), P, N, *grid_args(half, step),
                                                 order=order)
    assert int(cnt[0, 0]) == m == len(cc) and int(cnt[0, 1]) == int(cc[:, 2].sum())
    assert np.array_equal(idx[0].cpu().numpy(), ref_i), (half, step, P, N, order)
    assert np.array_equal(pil[0].cpu().numpy(), ref_p), (half, step, P, N, order, int(cc[:, 2].max()))
    # fused feature net on the same input
    torch.manual_seed(case)
    fn = M.PPFeatureNet(9, 64).to(gpu).eval()
    with torch.no_grad():
        fn.bn1.running_mean.normal_(0, 0.3)
        fn.bn1.running_var.uniform_(0.5, 1.5)
        fn.bn1.weight.normal_(0, 1.0)
        ref = fn(pil)
    feats, idx2 = vox.pfn(t, fn.fused_params())
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2)
    assert (feats - ref).abs().max().item() <= 1e-4
