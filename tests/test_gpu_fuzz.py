"""Randomised GPU parity sweep: many small (grid, P, N, density, order) draws,
dense output AND fused-feature-net output against the oracle / PyTorch.  Aims at
the rarely taken paths: LDS-pool overflow with medium buckets (zero pass + late
passes), ballot-rescan pillars, N not a multiple of 4, plane strides that are not
a multiple of 8 groups (group-granular deferral), rows longer than the line mask,
P smaller / larger than the number of occupied cells, both pillar orders."""
import os

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
        dense_hip = fn(pil)                 # pp_pfn_dense_dev (MFMA kernel when N % 4 == 0)
        fn.hip_eval = False
        ref = fn(pil)                       # PyTorch-ROCm
    feats, idx2 = vox.pfn(t, fn.fused_params())
    H = W = cfg.canvas_height
    canvas, idx3 = vox.pfn_canvas(t, fn.fused_params(), (H, W))
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2) and torch.equal(idx, idx3)
    assert (feats - ref).abs().max().item() <= 1e-4
    assert torch.equal(dense_hip, feats)    # same fmaf chain in both kernels
    sc = M.PPScatter(H, W)
    sc.channels_last_inference = False
    assert torch.equal(canvas, sc(feats, idx))


@pytest.mark.parametrize("case", list(range(16)))
def test_random_configuration_pipelined(gpu, oracle, case):
    """The same kind of draws through the software-pipelined form (k_step: ordered descriptors loaded beside the
    totals, the tile role's records fetched in pass 1, block order by launch size): a sequence of ragged batches
    whose shapes change from call to call, every result bit-equal to the plain call's and the first sweep of the
    first batch to the oracle."""
    import torch
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    rng = np.random.default_rng(5000 + case)
    half = float(rng.choice([4.0, 6.4, 10.0, 25.0]))
    step = float(rng.choice([0.2, 0.4, 0.5]))
    N = int(rng.choice([4, 6, 8, 12, 20, 32, 100, 260]))
    order = int(rng.integers(0, 2))
    nmax = int(rng.integers(800, 20000))
    first = _cloud(rng, nmax, half, int(rng.integers(0, 12)))
    cc = oracle.cell_counts(first.astype(np.float64), *grid_args(half, step))
    P = int(max(1, len(cc) * rng.choice([0.3, 0.9, 1.0, 1.5]) + rng.integers(0, 5)))   # P % 4 != 0 in most draws
    cfg = VoxelConfig.square(half, step, P, N, order=order)
    plain, piped = PillarVoxelizer(cfg, device=gpu), PillarVoxelizer(cfg, device=gpu)
    seq = []
    for i in range(int(rng.integers(3, 7))):
        B = int(rng.choice([1, 2, 5]))
        clouds = np.stack([first if (i == 0 and b == 0) else _cloud(rng, nmax, half, int(rng.integers(0, 12)))
                           for b in range(B)])
        ns = [nmax if (i == 0 and b == 0) else int(rng.integers(0, nmax + 1)) for b in range(B)]
        seq.append((torch.from_numpy(clouds).to(gpu), ns))
    want = [tuple(x.clone() for x in plain(t, n_points=ns, return_counts=True)) for t, ns in seq]
    got = [piped.submit(t, n_points=ns, return_counts=True) for t, ns in seq]
    got += [piped.submit(None, return_counts=True) for _ in range(piped.LAG)]
    got = [g for g in got if g is not None]
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for w_, g_ in zip(want, got):
        assert all(torch.equal(x, y) for x, y in zip(w_, g_)), (half, step, P, N, order)
    ref_p, ref_i, m = oracle.dataset_voxel_stage(first.astype(np.float64), P, N, *grid_args(half, step), order=order)
    assert np.array_equal(got[0][0][0].cpu().numpy(), ref_p) and np.array_equal(got[0][1][0].cpu().numpy(), ref_i)
    assert int(got[0][2][0, 0]) == m


@pytest.mark.parametrize("case", list(range(12)))
def test_random_configuration_fused_pipelined(gpu, case):
    """The fused feature-net form of the software pipeline (k_step<pfn>: emit role = conv1x1 + ReLU + BN + max +
    scatter, clear role for the other canvas) on random draws: ragged batches whose sizes change from call to call,
    both canvas layouts, dense submits mixed in -- every canvas bit-equal to the three-launch fused call's."""
    import torch
    import pp_amd.model as M
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    rng = np.random.default_rng(9000 + case)
    half = float(rng.choice([4.0, 6.4, 10.0, 25.0]))
    step = float(rng.choice([0.2, 0.4, 0.5]))
    N = int(rng.choice([4, 6, 8, 12, 20, 32, 100]))
    order = int(rng.integers(0, 2))
    nmax = int(rng.integers(800, 16000))
    P = int(rng.integers(50, 3000))
    cl = bool(rng.integers(0, 2))
    cfg = VoxelConfig.square(half, step, P, N, order=order)
    H = W = cfg.canvas_height
    torch.manual_seed(case)
    fn = M.PPFeatureNet(9, 64).to(gpu).eval()
    with torch.no_grad():
        fn.bn1.running_mean.normal_(0, 0.3)
        fn.bn1.running_var.uniform_(0.5, 1.5)
        fn.bn1.weight.normal_(0, 1.0)
    tab = fn.fused_params()
    plain, piped = PillarVoxelizer(cfg, device=gpu), PillarVoxelizer(cfg, device=gpu)
    seq = []
    for i in range(int(rng.integers(4, 8))):
        B = int(rng.choice([1, 2, 4]))
        clouds = np.stack([_cloud(rng, nmax, half, int(rng.integers(0, 12))) for _ in range(B)])
        ns = [int(rng.integers(0, nmax + 1)) for _ in range(B)]
        seq.append((torch.from_numpy(clouds).to(gpu), ns, bool(rng.integers(0, 4) == 0)))   # one call in four: dense
    want = []
    for t, ns, dense in seq:
        r = plain(t, n_points=ns) if dense else plain.pfn_canvas(t, tab, (H, W), n_points=ns, channels_last=cl)
        want.append(tuple(x.clone() for x in r))
    kinds = [d for _, _, d in seq]
    got = []
    calls = [(t, ns) for t, ns, _ in seq] + [(None, None)] * piped.LAG
    for k, (t, ns) in enumerate(calls):
        due = k - piped.LAG                                  # the batch this call emits decides the form it comes out in
        dense = kinds[due] if 0 <= due < len(kinds) else False
        r = piped.submit(t, n_points=ns) if dense else piped.submit_pfn_canvas(t, tab, (H, W), n_points=ns, channels_last=cl)
        if r is not None:
            got.append(tuple(x.clone() for x in r))
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for w_, g_ in zip(want, got):
        assert all(torch.equal(x, y) for x, y in zip(w_, g_)), (half, step, P, N, order, cl)


# PP_FUZZ_CASES=n in the environment adds n more seeds to the two target-assignment fuzz tests (the extra cases also draw
# 1-9 anchor types per cell: up to 8 stay on the box-centric kernel, 9 goes through the anchor-centric one)
_EXTRA = int(os.environ.get("PP_FUZZ_CASES", "0"))


@pytest.mark.parametrize("case", list(range(10 + _EXTRA)))
def test_random_target_assignment(gpu, oracle, case):
    """Random anchor grids (1-3 anchor types per cell, random sizes / yaws) and ground-truth sets
    (0-30 boxes, some duplicated, some far outside): anchors on the fly AND uploaded arrays
    against the oracle's create_target -- classes / flags exact, regression rows within 1e-6."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(7000 + case)
    fm = int(rng.integers(20, 70))
    per_cell = int(rng.integers(1, 4)) if case < 10 else int(rng.integers(1, 10))
    dims = tuple(tuple(float(v) for v in (rng.uniform(4, 14), rng.uniform(8, 30), rng.uniform(1, 3)))
                 for _ in range(per_cell))
    yaws = tuple(float(rng.choice([0.0, 90.0, 30.0])) for _ in range(per_cell))
    zs = tuple(float(rng.uniform(0.3, 1.2)) for _ in range(per_cell))
    acfg = boxes.AnchorConfig(fm, fm, 0.5, dims, yaws, zs)
    H = 2 * fm
    anchors = boxes.make_anchors(acfg)
    G = int(rng.integers(0, 31))
    gt = {"centers": np.column_stack([rng.uniform(-10, H + 10, G), rng.uniform(-10, H + 10, G), rng.uniform(0, 2, G)]),
          "wlh": np.column_stack([rng.uniform(4, 14, G), rng.uniform(8, 30, G), rng.uniform(1, 3, G)]),
          "yaw": rng.uniform(-np.pi, np.pi, G), "classes": rng.integers(0, 9, G).astype(np.int32)}
    if G >= 4:          # exact duplicates: ties in the column argmax, several classes on one anchor
        for k in ("centers", "wlh", "yaw"):
            gt[k][1] = gt[k][0]
        # a box that sits exactly on an anchor: IoU 1 with it
        d = int(rng.integers(0, per_cell))
        i = int(rng.integers(0, fm * fm)) * per_cell + d
        gt["centers"][2] = anchors["centers"][i]
        gt["wlh"][2] = anchors["wlh"][i]
        gt["yaw"][2] = anchors["yaw"][i]
    from pp_amd import boxes as _boxes   # same input arrays on both sides (see test_gpu_targets.py)
    # (no box: the oracle returns all-zero targets in its own shapes -- one code path for every G)
    c_img, k_img = _boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], H)
    ref_c, ref_r, _ = oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                           anchors["yaw"], gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], H,
                                           pos_thresh=0.6)
    for src in (acfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, device=gpu)
        cls_t, reg_t = ta.assign(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"], check=True)
        torch.cuda.synchronize()
        cls_t, reg_t = cls_t.cpu().numpy(), reg_t.cpu().numpy()
        assert np.array_equal(cls_t, ref_c.astype(np.float32)), (fm, per_cell, G)
        assert np.array_equal(reg_t[:, 0], ref_r[:, 0].astype(np.float32))
        assert np.array_equal(reg_t[:, 8], ref_r[:, 8].astype(np.float32))
        assert np.abs(reg_t - ref_r.astype(np.float32)).max() <= 1e-6


@pytest.mark.parametrize("case", list(range(12 + _EXTRA)))
def test_random_target_assignment_batches(gpu, oracle, case):
    """Random BATCHES through the one-launch form (pp_assign_targets[_grid]_batch_dev): 1-7 samples with 0-150 boxes
    each (so that a batch mixes the one-box-per-lane tail, the LDS tail with wave-specialised rows and samples of
    several 64-box chunks), 1-3 anchor types, feature-map scales that are and are not powers of two, 3-12 classes,
    clustered boxes (several clip rounds per workgroup), exact duplicates -- every sample against the oracle's
    create_target, anchors on the fly and uploaded."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    rng = np.random.default_rng(9100 + case)
    fm = int(rng.integers(24, 64))
    per_cell = int(rng.integers(1, 4)) if case < 12 else int(rng.integers(1, 10))
    scale = float(rng.choice([0.5, 0.5, 0.25, 0.4]))
    H = int(round(fm / scale))
    unit = H / (2.0 * fm)                                           # anchor spacing in canvas units / 2
    dims = tuple(tuple(float(v) for v in (rng.uniform(4, 14) * unit, rng.uniform(8, 30) * unit, rng.uniform(1, 3)))
                 for _ in range(per_cell))
    yaws = tuple(float(rng.choice([0.0, 90.0, 30.0])) for _ in range(per_cell))
    zs = tuple(float(rng.uniform(0.3, 1.2)) for _ in range(per_cell))
    acfg = boxes.AnchorConfig(fm, fm, scale, dims, yaws, zs)
    anchors = boxes.make_anchors(acfg)
    classes = int(rng.choice([3, 9, 12]))
    B = int(rng.integers(1, 8))
    gts = []
    for b in range(B):
        G = int(rng.choice([0, 1, 5, 30, 64, 65, 150]))
        c = np.column_stack([rng.uniform(-5, H + 5, G), rng.uniform(-5, H + 5, G), rng.uniform(0, 2, G)])
        if G >= 10:                                                  # a cluster: many pairs in a few workgroups
            n_cl = G // 2
            c[:n_cl, :2] = rng.uniform(0.2 * H, 0.8 * H, 2) + rng.uniform(-6, 6, (n_cl, 2))
        g = {"centers": c,
             "wlh": np.column_stack([rng.uniform(4, 14, G) * unit, rng.uniform(8, 30, G) * unit, rng.uniform(1, 3, G)]),
             "yaw": rng.uniform(-np.pi, np.pi, G), "classes": rng.integers(0, classes, G).astype(np.int32)}
        if G >= 4:
            for k in ("centers", "wlh", "yaw"):
                g[k][1] = g[k][0]                                    # an exact duplicate
            i = int(rng.integers(1, acfg.num_anchors))
            g["centers"][2], g["wlh"][2], g["yaw"][2] = anchors["centers"][i], anchors["wlh"][i], anchors["yaw"][i]
        gts.append(g)
    refs = []
    for g in gts:
        c_img, k_img = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], H)
        refs.append(oracle.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                         anchors["yaw"], g["centers"], g["wlh"], g["yaw"], g["classes"], H,
                                         pos_thresh=0.5, num_classes=classes)[:2])
    for src in (acfg, anchors):
        ta = TargetAssigner(src, canvas_height=H, pos_thresh=0.5, num_classes=classes, device=gpu)
        for _ in range(2):                                           # twice: every sample's scratch was re-armed
            cls_b, reg_b = ta.assign_batch(gts, check=True)
        torch.cuda.synchronize()
        cls_b, reg_b = cls_b.cpu().numpy(), reg_b.cpu().numpy()
        for b, (ref_c, ref_r) in enumerate(refs):
            tag = (case, fm, per_cell, scale, B, b, len(gts[b]["yaw"]))
            assert np.array_equal(cls_b[b], ref_c.astype(np.float32)), tag
            assert np.array_equal(reg_b[b][:, 0], ref_r[:, 0].astype(np.float32)), tag
            assert np.array_equal(reg_b[b][:, 8], ref_r[:, 8].astype(np.float32)), tag
            assert np.abs(reg_b[b] - ref_r.astype(np.float32)).max() <= 1e-6, tag
