"""GPU parity tests of the HIP voxelizer against the CPU oracle (through the C ABI).

Bar (BASELINE.json north_star): pillar indices / counts bit-exact; float features
within 1e-5.  The kernels are written to do better (same f64 operation order as
the oracle, one final f32 rounding), so most cases also assert bit equality.
"""
import numpy as np
import pytest

from util import C1, C2, C5, REFDEF, grid_args, oracle_stage, pillars_by_cell

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1e-5  # north_star tolerance on the float D-features


def _vox(gpu, half, step, P, N, order=0):
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    return PillarVoxelizer(VoxelConfig.square(half, step, P, N, order=order), device=gpu)


def _run(gpu, vox, pts32, n_points=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(pts32)).to(gpu)
    pil, idx, cnt = vox(t, n_points=n_points, return_counts=True)
    torch.cuda.synchronize()
    return pil.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()


def _check_exact(pil, idx, ref_p, ref_i):
    assert np.array_equal(idx, ref_i), "pillar indices differ"
    # occupancy pattern (which slots are non-zero) is integer work: exact
    assert np.array_equal(pil != 0, ref_p != 0)
    assert np.abs(pil - ref_p).max() <= FEATURE_TOL
    return np.array_equal(pil, ref_p)


@pytest.mark.parametrize("seed", [0, 1])
def test_below_cap_exact(gpu, oracle, seed):
    """cells <= P: exact set + content equality."""
    from pp_amd import synth
    pts = synth.lidar_like(20000, 20.0, seed)
    vox = _vox(gpu, 20.0, 0.2, 16000, 32)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, 16000, 32, 20.0, 0.2)
    assert m <= 16000 and cnt[0, 0] == m
    assert _check_exact(pil[0], idx[0], ref_p, ref_i), "features not bit-identical"


def test_overflow_same_order_and_subset_rule(gpu, oracle):
    """cells > P (the normal regime): identical to the oracle run in the same
    deterministic order, and every emitted pillar equals the UNCAPPED oracle's
    pillar for that cell whatever order the oracle used (SURVEY 7 hard part 1)."""
    from pp_amd import synth
    pts = synth.lidar_like(60000, 50.0, 3)
    P, N = 6000, 20
    vox = _vox(gpu, 50.0, 0.2, P, N)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, 50.0, 0.2)
    assert m > P and cnt[0, 0] == m
    assert (idx[0][:, 0] == 1).all(), "exactly P rows must be filled"
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)
    unc_p, unc_i, _ = oracle_stage(oracle, pts, m, N, 50.0, 0.2, order=oracle.ORDER_HASH)
    unc = pillars_by_cell(unc_p, unc_i)
    mine = pillars_by_cell(pil[0], idx[0])
    assert len(mine) == P
    for cell, block in mine.items():
        assert np.array_equal(block, unc[cell]), cell


def test_scrambled_order(gpu, oracle):
    from pp_amd import synth
    pts = synth.lidar_like(30000, 30.0, 5)
    P, N = 4000, 16
    vox = _vox(gpu, 30.0, 0.2, P, N, order=1)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, 30.0, 0.2, order=oracle.ORDER_SCRAMBLED)
    assert m > P
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)
    # the scrambled survivors are spread over the whole canvas, not the first rows
    rows = idx[0][:, 2]
    assert rows.max() - rows.min() > 0.8 * 300


def test_config1_dense_cells_big_pillar_path(gpu, oracle):
    """BASELINE config 1 (100x100 grid): hundreds of points per cell, i.e. the
    N-cap truncation and the ballot-rescan path for buckets beyond the LDS pool."""
    from pp_amd import synth
    c = C1
    pts = synth.lidar_like(c["n"], c["half"], 0)
    vox = _vox(gpu, c["half"], c["step"], c["P"], c["N"])
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, c["P"], c["N"], c["half"], c["step"])
    cc = oracle.cell_counts(pts.astype(np.float64), *grid_args(c["half"], c["step"]))
    assert cc[:, 2].max() > 128, "case must exercise the big-pillar path"
    assert cnt[0, 0] == m and cnt[0, 1] == cc[:, 2].sum()
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)


def test_all_points_in_one_cell(gpu, oracle):
    rng = np.random.default_rng(7)
    pts = np.zeros((5000, 4), np.float32)
    pts[:, 0] = 0.05 + 0.1 * rng.random(5000)
    pts[:, 1] = 0.05 + 0.1 * rng.random(5000)
    pts[:, 2] = rng.random(5000)
    pts[:, 3] = rng.random(5000)
    vox = _vox(gpu, 4.0, 0.2, 64, 100)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, 64, 100, 4.0, 0.2)
    assert m == 1 and cnt[0, 0] == 1 and cnt[0, 1] == 5000
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)


@pytest.mark.parametrize("P,N", [(1000, 7), (37, 10), (1, 1), (50, 13), (16, 64), (4095, 4)])
def test_odd_shapes(gpu, oracle, P, N):
    """N not a multiple of 4 (scalar store path), P not a multiple of the
    workgroup's pillar tile, tiny shapes."""
    from pp_amd import synth
    pts = synth.lidar_like(8000, 10.0, 11)
    vox = _vox(gpu, 10.0, 0.2, P, N)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, 10.0, 0.2)
    assert cnt[0, 0] == m
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)


def test_boundaries_nan_and_empty(gpu, oracle):
    """half-open box on all four sides, NaN/inf rows, empty cloud."""
    half, step = 2.0, 1.0
    nxt = np.nextafter
    f32 = np.float32
    rows = [
        [-2.0, 0.5, 0.0, 1],            # x == x_min      -> kept
        [nxt(f32(2.0), f32(0)), 0.5, 0, 2],   # just below x_max -> kept
        [2.0, 0.5, 0.0, 3],             # x == x_max      -> dropped
        [0.5, -2.0, 0.0, 4],            # y == y_min      -> kept
        [0.5, 2.0, 0.0, 5],             # y == y_max      -> dropped
        [0.5, 0.5, -10.0, 6],           # z == z_min      -> kept
        [0.5, 0.5, 10.0, 7],            # z == z_max      -> dropped
        [np.nan, 0.5, 0.0, 8],          # NaN             -> dropped (documented)
        [0.5, np.inf, 0.0, 9],
        [-0.0, -0.0, 0.0, 10],          # negative zero
    ]
    pts = np.array(rows, np.float32)
    vox = _vox(gpu, half, step, 16, 4)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, 16, 4, half, step)
    cc = oracle.cell_counts(pts.astype(np.float64), *grid_args(half, step))
    assert m == 4 and cc[:, 2].sum() == 5          # 4 cells, 5 surviving points
    assert cnt[0, 0] == m and cnt[0, 1] == cc[:, 2].sum()
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)
    # empty cloud and a cloud entirely out of range: all-zero outputs
    for cloud, npts in ((np.zeros((4, 4), np.float32), [0]),
                        (np.full((4, 4), 100.0, np.float32), None)):
        pil, idx, cnt = _run(gpu, vox, cloud, n_points=npts)
        assert not pil.any() and not idx.any() and cnt[0, 0] == 0 and cnt[0, 1] == 0


def test_batch_ragged_and_rerun_identical(gpu, oracle):
    """grid.y = sweep: a ragged batch equals per-sweep oracle runs; a second call
    on the same context (self-cleaned workspace) is bit-identical."""
    from pp_amd import synth
    ns = [9000, 12000, 1, 7000]
    clouds = [synth.lidar_like(12000, 15.0, 20 + i) for i in range(4)]
    batch = np.stack(clouds)
    P, N = 3000, 24
    vox = _vox(gpu, 15.0, 0.2, P, N)
    pil, idx, cnt = _run(gpu, vox, batch, n_points=ns)
    for b in range(4):
        ref_p, ref_i, m = oracle_stage(oracle, clouds[b][:ns[b]], P, N, 15.0, 0.2)
        assert cnt[b, 0] == m
        assert _check_exact(pil[b], idx[b], ref_p, ref_i), b
    pil2, idx2, cnt2 = _run(gpu, vox, batch, n_points=ns)
    assert np.array_equal(pil, pil2) and np.array_equal(idx, idx2) and np.array_equal(cnt, cnt2)


@pytest.mark.parametrize("order", [0, 1], ids=["row_major", "scrambled"])
@pytest.mark.parametrize("cfg", [C2, C5, REFDEF], ids=["config2", "config5", "reference_default"])
def test_full_size_configs(gpu, oracle, cfg, order):
    """BASELINE configs 2 and 5 and the reference's SHIPPED configuration (config.py:46-61,119-120: 600x600
    grid, P = 24000, N = 200 -- every bucket cap above the 128-point LDS pool, 172.8 MB per sweep) at full
    size, both pillar orders (scrambled is the default and what bench.py runs): oracle comparison plus the
    size-independent properties (exactly min(cells,P) rows, counts per cell, sortedness of the order, points
    conserved)."""
    from pp_amd import synth
    pts = synth.lidar_like(cfg["n"], cfg["half"], 0)
    P, N = cfg["P"], cfg["N"]
    vox = _vox(gpu, cfg["half"], cfg["step"], P, N, order=order)
    pil, idx, cnt = _run(gpu, vox, pts)
    pil, idx = pil[0], idx[0]
    cc = oracle.cell_counts(pts.astype(np.float64), *grid_args(cfg["half"], cfg["step"]))
    assert cnt[0, 0] == len(cc) and cnt[0, 1] == cc[:, 2].sum()
    nrow = min(len(cc), P)
    assert idx[:nrow, 0].all() and not idx[nrow:].any()
    if order == 0:
        # row-major order: (row, col) strictly increasing and equal to the first nrow cells
        assert np.array_equal(idx[:nrow, 1], cc[:nrow, 0]) and np.array_equal(idx[:nrow, 2], cc[:nrow, 1])
        key = idx[:nrow, 2] * 100000 + idx[:nrow, 1]
        assert (np.diff(key) > 0).all()
        want = cc[:nrow, 2]
    else:
        # scrambled order: ascending (cell * mult) mod ncells over the occupied cells
        nx = int(np.floor(2 * cfg["half"] / cfg["step"])) + 1
        H = int(round(2 * cfg["half"] / cfg["step"]))
        ncells = nx * nx
        mult = oracle.scramble_mult(ncells)
        # cell id = grid row from the top (canvas row + 1 guard row) * nx + col
        cell = (idx[:nrow, 2] + (nx - H)) * nx + idx[:nrow, 1]
        slot = (cell.astype(np.int64) * mult) % ncells
        assert (np.diff(slot) > 0).all()
        all_cell = (cc[:, 1] + (nx - H)) * nx + cc[:, 0]
        all_slot = np.sort((all_cell.astype(np.int64) * mult) % ncells)
        assert np.array_equal(slot, all_slot[:nrow])          # the first nrow occupied slots survive
        lut = {(int(c), int(r)): int(k) for c, r, k in cc}
        want = np.array([lut[(int(c), int(r))] for c, r in idx[:nrow, 1:]])
    # per-pillar live slot count == min(count, N); intensity channel (> 0 a.s.) marks live slots
    live = (pil[3] != 0).sum(axis=1)
    assert np.array_equal(live[:nrow], np.minimum(want, N))
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, cfg["half"], cfg["step"], order=order)
    assert _check_exact(pil, idx, ref_p, ref_i)
    # ... and the same cloud through the software-pipelined form (k_step: submit, then drain), against the
    # ORACLE's arrays directly, not only against the three-launch path
    import torch
    t = torch.from_numpy(pts).to(gpu)[None]
    outs = [vox.submit(t, return_counts=True)] + [vox.submit(None, return_counts=True) for _ in range(vox.LAG)]
    torch.cuda.synchronize()
    assert [o is None for o in outs] == [True] * vox.LAG + [False]
    sp, si, sc = (x.cpu().numpy() for x in outs[-1])
    assert np.array_equal(sc, cnt) and np.array_equal(sp[0], ref_p) and np.array_equal(si[0], ref_i)


def test_hip_graph_capture_and_replay(gpu, oracle):
    """The device entry point allocates and synchronises nothing once its workspace is
    reserved, so the four launches can be captured into a HIP graph and replayed; the
    self-cleaning workspace makes every replay independent of the previous one."""
    import torch
    from pp_amd import synth
    P, N = 3000, 16
    vox = _vox(gpu, 12.0, 0.2, P, N)
    a = torch.from_numpy(synth.lidar_like(9000, 12.0, 40)).to(gpu)
    b = torch.from_numpy(synth.lidar_like(9000, 12.0, 41)).to(gpu)
    static_in = a.clone().unsqueeze(0)
    out = (torch.empty((1, 9, P, N), dtype=torch.float32, device=gpu),
           torch.empty((1, P, 3), dtype=torch.int64, device=gpu))
    vox.reserve(1, 9000)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        vox(static_in, out=out)                      # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        vox(static_in, out=out)
    for cloud in (b, a, b):
        static_in.copy_(cloud.unsqueeze(0))
        g.replay()
        torch.cuda.synchronize()
        ref_p, ref_i, _ = oracle_stage(oracle, cloud.cpu().numpy(), P, N, 12.0, 0.2)
        assert np.array_equal(out[1][0].cpu().numpy(), ref_i)
        assert np.array_equal(out[0][0].cpu().numpy(), ref_p)


def test_soak_alternating_inputs_same_context(gpu):
    """300 back-to-back calls on ONE context, alternating clouds, batch sizes and the dense /
    fused entry points: every result must equal the first result for the same input -- the
    self-cleaning workspace (counts, scan words, tickets) and the look-back scan leave no
    state behind and have no ordering race."""
    import torch
    import pp_amd.model as M
    from pp_amd import synth
    P, N = 6000, 32
    vox = _vox(gpu, 25.0, 0.2, P, N)
    torch.manual_seed(0)
    prm = M.PPFeatureNet(9, 64).to(gpu).eval().fused_params()
    # same batch / capacity every call (the workspace layout never changes, so nothing is
    # re-zeroed by the host), different contents and ragged point counts
    clouds = [torch.from_numpy(np.stack([synth.lidar_like(30000, 25.0, 50 + 4 * c + b) for b in range(2)])).to(gpu)
              for c in range(3)]
    counts = [[30000, 30000], [12345, 30000], [30000, 1]]
    ref = {}
    for it in range(300):
        c = it % 3
        fused = (it // 3) % 2 == 1
        if fused:
            out, idx = vox.pfn(clouds[c], prm, n_points=counts[c])
        else:
            out, idx = vox(clouds[c], n_points=counts[c])
        key = (c, fused)
        sig = (out.double().sum().item(), out.abs().double().sum().item(), idx.sum().item())
        if key not in ref:
            ref[key] = (sig, out.clone(), idx.clone())
        else:
            assert sig == ref[key][0], (it, key)
            if it % 50 < 6:
                assert torch.equal(out, ref[key][1]) and torch.equal(idx, ref[key][2])
    torch.cuda.synchronize()


def test_data_mean_subtraction(gpu, oracle):
    """dataset.py:102-105: the optional per-element dataset mean (pillar_means.pkl) is
    subtracted from every sweep's [9,P,N] tensor, f32 - f32: bit-identical to the oracle's voxel
    stage with the same mean; odd element counts (scalar path); the fused modes refuse it."""
    import torch
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    for half, step, P, N in ((20.0, 0.2, 9000, 32), (10.0, 0.5, 333, 7)):
        rng = np.random.default_rng(3)
        mean = rng.normal(0, 2.0, 9 * P * N).astype(np.float32)
        cfg = VoxelConfig.square(half, step, P, N)
        vox = PillarVoxelizer(cfg, device=gpu, data_mean=mean)
        clouds = np.stack([synth.lidar_like(15000, half, s) for s in (0, 1)])
        pil, idx = vox(torch.from_numpy(clouds).to(gpu))
        torch.cuda.synchronize()
        for b in range(2):
            ref_p, ref_i, _ = oracle.dataset_voxel_stage(clouds[b].astype(np.float64), P, N,
                                                         *grid_args(half, step), data_mean=mean)
            assert np.array_equal(idx[b].cpu().numpy(), ref_i)
            assert np.array_equal(pil[b].cpu().numpy(), ref_p)
        with pytest.raises(RuntimeError):
            vox.pfn(torch.from_numpy(clouds).to(gpu), torch.zeros(64, 12, device=gpu))
        vox.set_data_mean(None)
        pil0, _ = vox(torch.from_numpy(clouds).to(gpu))
        assert torch.equal(pil0 - torch.from_numpy(mean).to(gpu).view(1, 9, P, N), pil)
    with pytest.raises(ValueError):
        PillarVoxelizer(cfg, device=gpu, data_mean=np.zeros(5, np.float32))


def test_check_reports_a_clean_stream(gpu, oracle):
    """pp_voxelize_check: synchronise + report.  The kernels have no failure mode of their own
    (no workgroup waits for another one), so after valid calls it reports nothing, however the
    calls were interleaved with other contexts' work."""
    import torch
    from pp_amd import synth
    pts = synth.lidar_like(20000, 20.0, 4)
    a = _vox(gpu, 20.0, 0.2, 6000, 32, order=1)
    b = _vox(gpu, 20.0, 0.2, 500, 8, order=0)
    ta = torch.from_numpy(pts).to(gpu)
    for _ in range(3):
        ra = a(ta, return_counts=True)
        rb = b(ta, return_counts=True)
    a.check()
    b.check()
    ref_p, ref_i, m = oracle_stage(oracle, pts, 6000, 32, 20.0, 0.2, order=1)
    assert int(ra[2][0, 0]) == m and _check_exact(ra[0][0].cpu().numpy(), ra[1][0].cpu().numpy(), ref_p, ref_i)
    ref_p, ref_i, m = oracle_stage(oracle, pts, 500, 8, 20.0, 0.2, order=0)
    assert int(rb[2][0, 0]) == m and _check_exact(rb[0][0].cpu().numpy(), rb[1][0].cpu().numpy(), ref_p, ref_i)


@pytest.mark.parametrize("order", [0, 1])
def test_more_than_one_window_of_chunks_in_a_tile(gpu, oracle, order):
    """300 000 points on a 9x9-cell grid: a single tile whose run list spans 293 split chunks,
    i.e. two windows of 256 in k_tile, all of it through the crowded-tile path (rounds fetched
    from the split arrays, nothing cached), buckets of thousands of points in k_emit."""
    rng = np.random.default_rng(42 + order)
    n = 300000
    pts = np.empty((n, 4), np.float32)
    pts[:, 0] = rng.uniform(-0.8, 0.8, n)
    pts[:, 1] = rng.uniform(-0.8, 0.8, n)
    pts[:, 2] = rng.uniform(-1, 1, n)
    pts[:, 3] = rng.uniform(0.1, 1, n)
    pts[::7, 0] = 5.0                      # out of range rows in between
    P, N = 40, 12
    vox = _vox(gpu, 0.8, 0.2, P, N, order=order)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, 0.8, 0.2, order=order)
    assert cnt[0, 0] == m == 64 and cnt[0, 1] == n - len(pts[::7])
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)


def test_maximum_batch(gpu, oracle):
    """PP_MAX_BATCH = 32 sweeps in one call (grid.y), ragged."""
    from pp_amd import synth
    B = 32
    clouds = np.stack([synth.lidar_like(3000, 8.0, 100 + i) for i in range(B)])
    ns = [3000 - 37 * i for i in range(B)]
    P, N = 700, 16
    vox = _vox(gpu, 8.0, 0.2, P, N, order=1)
    pil, idx, cnt = _run(gpu, vox, clouds, n_points=ns)
    for b in (0, 1, 15, 30, 31):
        ref_p, ref_i, m = oracle_stage(oracle, clouds[b][:ns[b]], P, N, 8.0, 0.2, order=1)
        assert cnt[b, 0] == m
        assert _check_exact(pil[b], idx[b], ref_p, ref_i), b
    with pytest.raises(ValueError):
        import torch
        vox(torch.zeros((33, 8, 4), device=gpu))


@pytest.mark.parametrize("order", [0, 1])
def test_largest_grids(gpu, oracle, order):
    """4001 x 4001 cells = 3909 tiles of 4096 slots: the split's four bins per thread, tiles
    with 64 cells per k_tile thread, 62 tiles per lane in k_emit's prefix.  One more doubling of
    the grid is beyond the documented limit and must be refused."""
    import torch
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    pts = synth.lidar_like(30000, 400.0, 9)
    pts[:2000] = synth.lidar_like(2000, 3.0, 10)          # and a crowded spot
    P, N = 20000, 8
    vox = _vox(gpu, 400.0, 0.2, P, N, order=order)
    pil, idx, cnt = _run(gpu, vox, pts)
    ref_p, ref_i, m = oracle_stage(oracle, pts, P, N, 400.0, 0.2, order=order)
    assert cnt[0, 0] == m
    assert _check_exact(pil[0], idx[0], ref_p, ref_i)
    too_big = PillarVoxelizer(VoxelConfig.square(450.0, 0.2, 100, 4, order=order), device=gpu)   # 4501^2 > 2^24
    with pytest.raises(ValueError, match="too large"):
        too_big(torch.from_numpy(pts[:100]).to(gpu))


def test_pipelined_mode_equals_plain_calls(gpu, oracle):
    """PillarVoxelizer.submit / pp_voxelize_step_dev (ONE launch per call: the split stage of batch i, the
    tile stage of batch i-1, the order stage of batch i-2 and the emit stage of batch i-3 as roles of one grid): a sequence of DIFFERENT
    batches -- changing batch size, ragged row counts, an empty sweep -- comes out in order and bit-identical
    to plain calls on another context; the first batch also against the oracle.  Plain calls on the same
    context in between do not disturb the batches in flight."""
    import torch
    from pp_amd import synth
    half, step, P, N = 25.0, 0.25, 3000, 24
    vp, vs = _vox(gpu, half, step, P, N, order=1), _vox(gpu, half, step, P, N, order=1)
    seq = []
    for i, (B, n) in enumerate([(2, 9000), (1, 20000), (4, 5000), (2, 9000), (3, 12000), (1, 300)]):
        pts = np.stack([synth.lidar_like(n, half, 50 + 7 * i + s) for s in range(B)])
        npts = [n - 37 * s for s in range(B)]
        if i == 2:
            npts[1] = 0
        seq.append((torch.from_numpy(pts).to(gpu), npts))
    plain = []
    for t, npts in seq:
        p, ix, c = vp(t, n_points=npts, return_counts=True)
        plain.append((p.clone(), ix.clone(), c.clone()))
    got = []
    for rep in range(2):                      # twice: the slots are reused with stale contents
        got = []
        for k, (t, npts) in enumerate(seq):
            r = vs.submit(t, n_points=npts, return_counts=True)
            assert (r is None) == (k < vs.LAG)
            if r is not None:
                got.append(r)
            if k == 3:                        # a plain call on the context with batches in flight
                assert torch.equal(vs(seq[1][0], n_points=seq[1][1])[0], plain[1][0])
        for _ in range(vs.LAG):
            got.append(vs.submit(None, return_counts=True))
        assert vs.submit(None) is None        # drained
        torch.cuda.synchronize()
        assert len(got) == len(seq)
        for (p, ix, c), (q, jx, d) in zip(plain, got):
            assert torch.equal(p, q) and torch.equal(ix, jx) and torch.equal(c, d)
    ref_p, ref_i, m = oracle_stage(oracle, seq[0][0][0, :seq[0][1][0]].cpu().numpy(), P, N, half, step, order=1)
    assert np.array_equal(got[0][0][0].cpu().numpy(), ref_p) and np.array_equal(got[0][1][0].cpu().numpy(), ref_i)
    # the generator form
    outs = list(vs.stream([t for t, _ in seq[:4]]))
    torch.cuda.synchronize()
    assert len(outs) == 4 and all(torch.equal(o[0], vp(t)[0]) for o, (t, _) in zip(outs, seq[:4]))


@pytest.mark.parametrize("order", [0, 1], ids=["row_major", "scrambled"])
@pytest.mark.parametrize("cfg", [C2, C5, C1, REFDEF], ids=["config2", "config5", "config1", "reference_default"])
def test_pipelined_mode_full_size(gpu, cfg, order):
    """k_step at BASELINE's sizes (and configs[0]'s crowded cells): bit-identical to the three-kernel path,
    odd N (scalar store mode) included."""
    import torch
    from pp_amd import synth
    for N in (cfg["N"], 37):
        a, b = _vox(gpu, cfg["half"], cfg["step"], cfg["P"], N, order), _vox(gpu, cfg["half"], cfg["step"], cfg["P"], N, order)
        ts = [torch.from_numpy(np.stack([synth.lidar_like(cfg["n"], cfg["half"], 10 * i + s) for s in range(2)])).to(gpu)
              for i in range(3)]
        outs = list(b.stream(ts))
        for t, o in zip(ts, outs):
            p, ix = a(t)
            assert torch.equal(p, o[0]) and torch.equal(ix, o[1])


def test_pipelined_mode_edge_shapes(gpu):
    """k_step at the edges: PP_MAX_BATCH = 32 ragged sweeps per call, the largest grids (3909 tiles of 4096
    slots: the split role's 16 bins per thread, 47 KB of dynamic LDS), a batch with no point at all, a grid
    beyond the limit refused -- every result bit-identical to the three-launch form."""
    import torch
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    # 32 ragged sweeps
    B = 32
    clouds = torch.from_numpy(np.stack([synth.lidar_like(3000, 8.0, 100 + i) for i in range(B)])).to(gpu)
    ns = [3000 - 37 * i for i in range(B)]
    a, b = _vox(gpu, 8.0, 0.2, 700, 16, order=1), _vox(gpu, 8.0, 0.2, 700, 16, order=1)
    want = a(clouds, n_points=ns, return_counts=True)
    outs = [b.submit(clouds, n_points=ns, return_counts=True) for _ in range(2)] + \
           [b.submit(None, return_counts=True) for _ in range(b.LAG)]
    got = [o for o in outs if o is not None]
    assert len(got) == 2 and all(torch.equal(x, y) for g in got for x, y in zip(g, want))
    # largest grids, both orders, with a crowded spot; then nothing but out-of-range points
    pts = synth.lidar_like(30000, 400.0, 9)
    pts[:2000] = synth.lidar_like(2000, 3.0, 10)
    t = torch.from_numpy(pts).to(gpu)[None]
    far = torch.full((1, 500, 4), 1.0e6, device=gpu)
    for order in (0, 1):
        a, b = _vox(gpu, 400.0, 0.2, 20000, 8, order=order), _vox(gpu, 400.0, 0.2, 20000, 8, order=order)
        seq = [t, far, t]
        outs = list(b.stream(seq))
        for x, o in zip(seq, outs):
            p, ix = a(x)
            assert torch.equal(p, o[0]) and torch.equal(ix, o[1])
        assert not outs[1][0].any() and not outs[1][1].any()
    # an abandoned stream: reset, then the pipeline starts empty again
    b.submit(t), b.submit(far)
    b.reset_stream()
    outs = list(b.stream([t]))
    assert len(outs) == 1 and torch.equal(outs[0][0], a(t)[0])
    too_big = PillarVoxelizer(VoxelConfig.square(450.0, 0.2, 100, 4), device=gpu)   # 4501^2 > 2^24 cells
    with pytest.raises(ValueError, match="too large"):
        too_big.submit(t)
    assert too_big.submit(None) is None       # the refused batch left nothing in flight


def test_pipelined_mode_soak(gpu):
    """Three hundred back-to-back submits cycling through inputs of different sizes and batch shapes (so that the
    four workspace slots are re-laid-out while older batches are still in flight), cache-flushing traffic every
    few calls: every returned batch equals the plain call's result for ITS input."""
    import torch
    from pp_amd import synth
    half, step, P, N = 30.0, 0.25, 5000, 32
    plain, vs = _vox(gpu, half, step, P, N, order=1), _vox(gpu, half, step, P, N, order=1)
    inputs = []
    for i, (B, n) in enumerate([(1, 30000), (3, 8000), (2, 15000), (4, 4000), (1, 500)]):
        inputs.append(torch.from_numpy(np.stack([synth.lidar_like(n, half, 900 + 10 * i + s) for s in range(B)])).to(gpu))
    want = [tuple(x.clone() for x in plain(t)) for t in inputs]
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=gpu)
    order, got = [], []
    for k in range(300):                      # no host synchronisation inside: the launches queue up back to back
        i = (k * 7 + k // 5) % len(inputs)
        order.append(i)
        r = vs.submit(inputs[i])
        if k % 11 == 0:
            junk.zero_()
        if r is not None:
            got.append(r)
    got += [vs.submit(None) for _ in range(vs.LAG)]
    torch.cuda.synchronize()
    assert len(got) == 300
    bad = [k for k, (r, i) in enumerate(zip(got, order))
           if not (torch.equal(r[0], want[i][0]) and torch.equal(r[1], want[i][1]))]
    assert not bad, bad[:10]


def test_pipelined_mode_stream_is_enforced_and_refused_submit_changes_nothing(gpu):
    """pp_voxelize_step_dev carries a batch from role to role by STREAM ORDER: a submit on another stream while
    batches are in flight is refused (PP_ERR_VALUE, "another stream") instead of racing; once drained, any stream
    may start the next pipeline.  A refusal -- that one, or a bad argument in the middle of a full pipeline (here:
    more sweeps than PP_MAX_BATCH) -- changes NOTHING on either side (include/pp_hip.h): the batches in
    flight come out, in order, on their own stream."""
    import torch
    from pp_amd import synth
    half, step, P, N = 20.0, 0.25, 2000, 16
    a, b = _vox(gpu, half, step, P, N, order=1), _vox(gpu, half, step, P, N, order=1)
    ts = [torch.from_numpy(synth.lidar_like(6000, half, 70 + i)).to(gpu)[None] for i in range(3)]
    want = [tuple(x.clone() for x in a(t)) for t in ts]
    other = torch.cuda.Stream(device=gpu)
    assert b.submit(ts[0]) is None and b.submit(ts[1]) is None
    torch.cuda.synchronize()
    with torch.cuda.stream(other):
        with pytest.raises(ValueError, match="another stream"):
            b.submit(ts[2])
    # the two batches are still in flight: they come out on their own stream, then the pipeline is empty
    got = [b.submit(None) for _ in range(b.LAG)]
    torch.cuda.synchronize()
    assert got[0] is None and all(torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) for g, w in zip(got[1:], want))
    # drained: the other stream may start the next pipeline
    with torch.cuda.stream(other):
        outs = list(b.stream(ts))
    other.synchronize()
    assert len(outs) == 3 and all(torch.equal(o[0], w[0]) and torch.equal(o[1], w[1]) for o, w in zip(outs, want))
    # C ABI directly: the refusal itself changes nothing -- the batches in flight come out on their own stream
    import ctypes
    from pp_amd import _lib
    c = _vox(gpu, half, step, P, N, order=1)
    assert c.submit(ts[0]) is None
    s_other = ctypes.c_void_p(other.cuda_stream)
    em = ctypes.c_int(0)
    n1 = (ctypes.c_int32 * 1)(6000)
    rc = _lib.lib().pp_voxelize_step_dev(c._ctx.handle, s_other, ctypes.c_void_p(ts[1].data_ptr()), 6000, n1, 1,
                                         ctypes.byref(c._prm), None, None, None, ctypes.byref(em))
    assert rc == _lib.PP_ERR_VALUE and b"another stream" in _lib.lib().pp_last_error()
    got = [c.submit(None) for _ in range(c.LAG)]
    torch.cuda.synchronize()
    assert got[:2] == [None, None] and torch.equal(got[2][0], want[0][0]) and torch.equal(got[2][1], want[0][1])
    # a rejected submit in the middle of a FULL pipeline
    d = _vox(gpu, half, step, P, N, order=1)
    for t in ts:
        assert d.submit(t) is None
    too_many = torch.zeros((33, 16, 4), device=gpu)                 # PP_MAX_BATCH is 32
    with pytest.raises(ValueError):
        d.submit(too_many)
    # the three good batches are still in flight, on both sides: they drain in order
    outs = [d.submit(None) for _ in range(d.LAG)]
    torch.cuda.synchronize()
    assert all(o is not None and torch.equal(o[0], w[0]) and torch.equal(o[1], w[1]) for o, w in zip(outs, want))
    assert d.submit(None) is None                                    # ... and the pipeline is empty
    outs = list(d.stream(ts[:2]))
    torch.cuda.synchronize()
    assert len(outs) == 2 and torch.equal(outs[0][0], want[0][0]) and torch.equal(outs[1][1], want[1][1])


def test_timing_ring_does_not_mix_launch_kinds(gpu):
    """pp_ctx_read_kernel_ms after plain calls FOLLOWING pipelined calls (and the other way round) without a
    set_timing in between: only the entries of the current kind are reported -- no stale or never-recorded
    event pairs, no k_emit durations among k_step's."""
    import torch
    from pp_amd import _lib, synth
    half, step, P, N = 20.0, 0.25, 2000, 16
    v = _vox(gpu, half, step, P, N, order=1)
    t = torch.from_numpy(synth.lidar_like(6000, half, 3)).to(gpu)[None]
    v.set_timing(8)
    for _ in range(5):
        v.submit(t)                                   # five k_step launches
    for _ in range(2):
        v(t)                                          # then two three-launch calls
    torch.cuda.synchronize()
    sp, tl, em = (v.read_kernel_ms(w) for w in (_lib.KERNEL_SPLIT, _lib.KERNEL_TILE, _lib.KERNEL_EMIT))
    assert len(sp) == len(tl) == len(em) == 2 and all(0 < x < 50 for x in sp + tl + em)
    for _ in range(3):
        v(t)
    for _ in range(4):
        v.submit(t)
    torch.cuda.synchronize()
    assert v.read_kernel_ms(_lib.KERNEL_SPLIT) == [] and len(v.read_kernel_ms(_lib.KERNEL_EMIT)) == 4
    v.set_timing(0)
    v.reset_stream()


def test_pipelined_mode_split_into_sub_batch_launches(gpu):
    """pp_voxelize_step_dev can send a call out as several launches of a few sweeps each (development knob
    PP_STEP_SUB_MB; off by default: measured, no gain).  The roles' sweep offsets must then address the same
    arrays: in a child process with the knob set so that a 3-sweep batch takes three launches and a 2-sweep batch
    two, every result equals the plain call's, ragged counts and an empty sweep included."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import pp_amd
from pp_amd import synth
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
cfg = VoxelConfig.square(25.0, 0.25, 3000, 64)          # 36 * 3000 * 64 = 6.9 MB per sweep, knob = 8 MB: one sweep per launch
a, b = PillarVoxelizer(cfg), PillarVoxelizer(cfg)
seq = []
for i, (B, n) in enumerate([(3, 9000), (1, 20000), (4, 5000), (2, 9000), (3, 12000)]):
    pts = torch.from_numpy(np.stack([synth.lidar_like(n, 25.0, 50 + 7 * i + s) for s in range(B)])).cuda()
    npts = [n - 37 * s for s in range(B)]
    if i == 2:
        npts[1] = 0
    seq.append((pts, npts))
want = [tuple(x.clone() for x in a(t, n_points=n, return_counts=True)) for t, n in seq]
got = []
for t, n in seq:
    r = b.submit(t, n_points=n, return_counts=True)
    if r is not None:
        got.append(r)
got += [b.submit(None, return_counts=True) for _ in range(b.LAG)]
torch.cuda.synchronize()
assert len(got) == len(want)
for w, g in zip(want, got):
    assert all(torch.equal(x, y) for x, y in zip(w, g))
print("sub-batch launches ok")
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, PP_STEP_SUB_MB="8"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sub-batch launches ok" in r.stdout, r.stderr[-2000:]


# --------------------------------------------------------------------------- grids that are not square
def _rect_cloud(n, seed):
    """Points around an off-centre sensor: most inside x in [-30, 50), y in [-12, 20), some outside on every side,
    a dense patch (cells far beyond N points) and points exactly on cell and range boundaries."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 4), np.float32)
    pts[:, 0] = rng.normal(8.0, 22.0, n)
    pts[:, 1] = rng.normal(3.0, 9.0, n)
    pts[:, 2] = rng.uniform(-3.0, 3.0, n)
    pts[:, 3] = rng.uniform(0.0, 255.0, n)
    k = n // 10
    pts[:k, 0] = rng.uniform(11.0, 12.5, k)                      # the dense patch
    pts[:k, 1] = rng.uniform(-2.0, -0.4, k)
    edge = np.array([[-30.0, -12.0], [50.0, 20.0], [-30.0, 19.999998], [49.999996, -12.0], [0.25, 0.4], [10.0, 8.0],
                     [-29.75, -11.6], [49.75, 19.6]], np.float32)
    pts[k:k + len(edge), :2] = edge
    return pts


@pytest.mark.parametrize("canvas_height", [80, 200, 33])       # the grid's row count, a larger canvas, a smaller one
@pytest.mark.parametrize("order", [0, 1])
def test_rectangular_offset_grid_device_api(gpu, oracle, order, canvas_height):
    """create_pillars takes nine independent grid scalars (pillars.cpp:236-249): x_step = 0.25, y_step = 0.4,
    x in [-30, 50), y in [-12, 20) -- 320 x 80 cells, not centred -- and a canvas_height that is, is larger than, and is
    smaller than the row count (the reference computes (canvas_height - 1) - floor((y - y_min) / y_step),
    pillars.cpp:278-280: negative rows included).  Through the plain call AND the software-pipelined submit, with and
    without overflow, both orders: indices exact, features bit for bit the oracle's."""
    import torch
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    pts = _rect_cloud(50000, 4 + canvas_height)
    dev = torch.from_numpy(pts).to(gpu)
    for P, N in ((20000, 24), (2500, 24), (7000, 3)):            # no overflow / overflow / tiny N (every busy cell truncated)
        cfg = VoxelConfig.rect((-30.0, 50.0), (-12.0, 20.0), 0.25, 0.4, P, N, z_range=(-2.5, 2.5),
                               canvas_height=canvas_height, order=order)
        assert cfg.canvas_width == 320
        ref_p, ref_i, m = oracle.dataset_voxel_stage(pts.astype(np.float64), P, N, *cfg.grid_args(), order=order)
        assert (m > P) == (P in (2500, 7000))
        if canvas_height == 33:
            assert (ref_i[:, 2] < 0).any()                       # rows above the canvas come out negative, as in the reference
        vox = PillarVoxelizer(cfg, device=gpu)
        pil, idx, cnt = vox(dev, return_counts=True)
        torch.cuda.synchronize()
        assert int(cnt[0, 0]) == m
        assert np.array_equal(idx[0].cpu().numpy(), ref_i), (P, N, "indices")
        assert np.array_equal(pil[0].cpu().numpy(), ref_p), (P, N, "features")
        outs = list(vox.stream([dev, dev[:30000].contiguous(), dev]))
        torch.cuda.synchronize()
        assert len(outs) == 3
        for k in (0, 2):
            assert torch.equal(outs[k][0], pil) and torch.equal(outs[k][1], idx), (P, N, "pipelined", k)
        ref_p2, ref_i2, _ = oracle.dataset_voxel_stage(pts[:30000].astype(np.float64), P, N, *cfg.grid_args(), order=order)
        assert np.array_equal(outs[1][1][0].cpu().numpy(), ref_i2) and np.array_equal(outs[1][0][0].cpu().numpy(), ref_p2)


def test_grid_beyond_32768_cells_per_axis_is_refused(gpu):
    """The library packs a cell's column and row into 15 bits each: a grid beyond 32 768 cells on an axis is refused
    with ValueError and a message that names the limit (a documented tightening: the reference has no limit, its hash
    map is keyed on doubles) -- on the plain call, on submit (the pipeline stays usable) and on the host module."""
    import torch
    from pp_amd import pillars
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    pts = torch.zeros((64, 4), dtype=torch.float32, device=gpu)
    wide = VoxelConfig.rect((0.0, 3300.0), (0.0, 10.0), 0.1, 0.1, 100, 4)        # 33 000 columns
    vox = PillarVoxelizer(wide, device=gpu)
    with pytest.raises(ValueError, match="limit 32768 per axis"):
        vox(pts)
    with pytest.raises(ValueError, match="limit 32768 per axis"):
        vox.submit(pts)
    tall = VoxelConfig.rect((0.0, 10.0), (0.0, 3300.0), 0.1, 0.1, 100, 4)
    with pytest.raises(ValueError, match="limit 32768 per axis"):
        PillarVoxelizer(tall, device=gpu)(pts)
    with pytest.raises(ValueError, match="limit 32768 per axis"):
        pillars.create_pillars(np.zeros((8, 4)), np.zeros((100, 4, 9)), np.zeros((100, 3)), 4, 100,
                               *wide.grid_args())
    ok = VoxelConfig.rect((0.0, 3276.0), (0.0, 10.0), 0.1, 0.1, 100, 4)          # 32 761 columns: accepted
    p, i = PillarVoxelizer(ok, device=gpu)(pts)
    torch.cuda.synchronize()
    assert int(i[0, 0, 0]) == 1 and int(i[0, 1, 0]) == 0                         # 64 points at the origin: one pillar


@pytest.mark.parametrize("N", [4, 100, 200])
@pytest.mark.parametrize("order", [0, 1])
def test_crowded_cells_at_every_slice_boundary(gpu, oracle, order, N):
    """The streamed path of a wave whose buckets overflow its LDS pool (streamed_means + emit_live_run): cells with
    exactly 1, 3, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 130, 255, 256, 257, 381 and 700 points, laid out so that
    the four pillars of a wave mix every combination of short, pool-sized and streamed buckets (in row-major order
    the counts follow each other cell by cell; the scrambled order mixes them differently), N below, at and above the
    pool (4 / 100 / 200: live points in one run, several runs, more than the pool holds), points shuffled: the plain
    call, the software-pipelined call and -- through PPFeatureNet's fused form -- the pooled-maximum path all bit
    for bit against the oracle (the running mean is a sequential f64 chain: pillars.cpp:311-328)."""
    import torch
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    rng = np.random.default_rng(1000 * N + order)
    counts = [1, 3, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 130, 255, 256, 257, 381, 700]
    W = H = 12
    cell_counts = [counts[(i * 7 + (i // W)) % len(counts)] if (i % 5) else 0 for i in range(W * H)]   # some cells empty
    pts = []
    for i, c in enumerate(cell_counts):
        cx, cy = i % W, i // W
        p = np.empty((c, 4), np.float32)
        p[:, 0] = cx + rng.uniform(0.01, 0.99, c)
        p[:, 1] = cy + rng.uniform(0.01, 0.99, c)
        p[:, 2] = rng.uniform(-1.0, 1.0, c)
        p[:, 3] = rng.uniform(0, 255, c)
        pts.append(p)
    pts = np.concatenate(pts)
    rng.shuffle(pts)
    P = 160
    cfg = VoxelConfig.rect((0.0, float(W)), (0.0, float(H)), 1.0, 1.0, P, N, z_range=(-2.0, 2.0), order=order)
    ref_p, ref_i, m = oracle.dataset_voxel_stage(pts.astype(np.float64), P, N, *cfg.grid_args(), order=order)
    assert m == sum(1 for c in cell_counts if c) and m <= P
    dev = torch.from_numpy(pts).to(gpu)
    vox = PillarVoxelizer(cfg, device=gpu)
    pil, idx = vox(dev)
    torch.cuda.synchronize()
    assert np.array_equal(idx[0].cpu().numpy(), ref_i)
    assert np.array_equal(pil[0].cpu().numpy(), ref_p)
    outs = list(vox.stream([dev, dev]))
    torch.cuda.synchronize()
    assert all(torch.equal(o[0], pil) and torch.equal(o[1], idx) for o in outs)
    # two sweeps per launch, the second with a different shuffle: same pillars, rows in the other input order
    pts2 = pts.copy()
    rng.shuffle(pts2)
    ref_p2, ref_i2, _ = oracle.dataset_voxel_stage(pts2.astype(np.float64), P, N, *cfg.grid_args(), order=order)
    both = torch.from_numpy(np.stack([pts, pts2])).to(gpu)
    pb, ib = vox(both)
    torch.cuda.synchronize()
    assert np.array_equal(pb[0].cpu().numpy(), ref_p) and np.array_equal(pb[1].cpu().numpy(), ref_p2)
    assert np.array_equal(ib[1].cpu().numpy(), ref_i2)
    # the fused feature net takes the same path with a 64-point pool (slices of 64 / 32 / 16): its features must be the
    # bits of the feature-net kernel run on the dense tensor above (same arithmetic: tests/test_gpu_pfn.py)
    from test_gpu_pfn import _net
    fn = _net(gpu)
    with torch.no_grad():
        fn.hip_eval = True
        want = fn(pb)
    feats, idx_f = vox.pfn(both, fn.fused_params())
    torch.cuda.synchronize()
    assert torch.equal(idx_f, ib) and torch.equal(feats, want)
