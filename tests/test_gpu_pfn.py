"""GPU parity of the fused voxelizer + feature net against
PPFeatureNet (model/model.py:31-40) evaluated by PyTorch on the dense tensor.

Bar: both are f32 evaluations of the same function (the fused kernel sums the
9-term dot product in one fixed fmaf order and applies eval-mode BatchNorm as
s*r+t; MIOpen uses its own order), so each is compared with an f64 evaluation of
PPFeatureNet: the fused kernel's error must not exceed twice PyTorch-f32's own
(+1e-6), and the two f32 results must agree within 1e-4 absolute (inputs reach
|xp|,|yp| ~ 500, outputs ~ 60)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_ABS = 1e-4


def _net(gpu, seed=0):
    import torch
    import pp_amd.model as M
    torch.manual_seed(seed)
    fn = M.PPFeatureNet(9, 64).to(gpu)
    with torch.no_grad():      # non-trivial BN statistics, both signs of the BN scale
        fn.bn1.running_mean.normal_(0, 0.5)
        fn.bn1.running_var.uniform_(0.3, 2.0)
        fn.bn1.weight.normal_(0, 1.0)
        fn.bn1.bias.normal_(0, 0.3)
        fn.conv1.weight.mul_(0.2)
    return fn.eval()


def _compare(gpu, cfg_args, pts_list, n_points=None, order=0):
    import torch
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    half, step, P, N = cfg_args
    vox = PillarVoxelizer(VoxelConfig.square(half, step, P, N, order=order), device=gpu)
    fn = _net(gpu)
    pts = torch.from_numpy(np.stack(pts_list)).to(gpu)
    dense, idx = vox(pts, n_points=n_points)
    import pp_amd.model as M
    fn64 = M.PPFeatureNet(9, 64).to(gpu).double()
    fn64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in fn.state_dict().items()})
    fn64.eval()
    with torch.no_grad():
        fn.hip_eval = False
        ref = fn(dense)                                    # [B,64,P], PyTorch-ROCm f32
        fn.hip_eval = True
        hip = fn(dense)                                    # pp_pfn_dense_dev on the dense tensor
        ref64 = fn64(dense.double())
    feats, idx2, cnt = vox.pfn(pts, fn.fused_params(), n_points=n_points, return_counts=True)
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2)
    # the dense-tensor kernel and the fused voxelizer run the same arithmetic: identical bits
    assert torch.equal(hip, feats)
    assert (fn.bn1.weight < 0).any() and (fn.bn1.weight > 0).any()
    err_torch = (ref.double() - ref64).abs().max().item()
    err_fused = (feats.double() - ref64).abs().max().item()
    assert err_fused <= 2 * err_torch + 1e-6, (err_fused, err_torch)
    assert (feats - ref).abs().max().item() <= TOL_ABS
    return feats, ref


def test_fused_pfn_matches_torch_feature_net(gpu):
    from pp_amd import synth
    _compare(gpu, (20.0, 0.2, 8000, 32), [synth.lidar_like(20000, 20.0, s) for s in (0, 1)])


def test_fused_pfn_overflow_empty_rows_and_ragged_batch(gpu):
    from pp_amd import synth
    clouds = [synth.lidar_like(30000, 30.0, 5 + s) for s in range(3)]
    feats, ref = _compare(gpu, (30.0, 0.2, 5000, 16), clouds, n_points=[30000, 1200, 0])
    # sweep 2 is empty: every row equals BN(ReLU(bias)) -- the zero-padded value
    assert (feats[2] == feats[2][:, :1]).all()


def test_fused_pfn_dense_cells_ncap_and_big_pillars(gpu):
    """BASELINE config 1 grid: up to 381 points per cell, N-cap at 100 (no padding in
    those pillars), the ballot-rescan path, and P not a multiple of 4."""
    from pp_amd import synth
    _compare(gpu, (50.0, 1.0, 5763, 100), [synth.lidar_like(60000, 50.0, 0)])
    _compare(gpu, (50.0, 1.0, 3001, 8), [synth.lidar_like(60000, 50.0, 1)], order=1)


def test_fused_pipeline_equals_dense_pipeline(gpu):
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    pipe = PillarPipeline(VoxelConfig.square(16.0, 0.2, 4000, 32), feature_channels=64, device=gpu, seed=0)
    pipe.model.eval()
    pts = torch.from_numpy(np.stack([synth.lidar_like(15000, 16.0, s) for s in (3, 4)])).to(gpu)
    c1, r1 = pipe.forward(pts)
    c2, r2 = pipe.forward_fused(pts)
    torch.cuda.synchronize()
    assert (c1 - c2).abs().max().item() <= 1e-4 * max(1.0, c1.abs().max().item())
    assert (r1 - r2).abs().max().item() <= 1e-4 * max(1.0, r1.abs().max().item())
    pipe.model.train()
    with pytest.raises(RuntimeError):
        pipe.forward_fused(pts)


def test_fused_scatter_canvas_equals_ppscatter(gpu):
    """pp_voxelize_pfn_canvas_dev == PPScatter(PPFeatureNet(...)) (model/model.py:53-62), bit for
    bit against the unfused pfn output scattered by PyTorch: both memory layouts, overflow
    (cells > P: the dropped cells stay zero), a ragged batch with an empty sweep."""
    import torch
    import pp_amd.model as M
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    half, step, P, N = 24.0, 0.2, 3000, 16
    vox = PillarVoxelizer(VoxelConfig.square(half, step, P, N), device=gpu)
    H = W = vox.cfg.canvas_height
    fn = _net(gpu)
    pts = torch.from_numpy(np.stack([synth.lidar_like(20000, half, s) for s in (1, 2, 3)])).to(gpu)
    n_points = [20000, 700, 0]
    feats, idx, cnt = vox.pfn(pts, fn.fused_params(), n_points=n_points, return_counts=True)
    assert cnt[0, 0].item() > P          # overflow regime on sweep 0
    sc = M.PPScatter(H, W)
    sc.channels_last_inference = False
    want = sc(feats, idx)
    for cl in (True, False):
        canvas, idx2 = vox.pfn_canvas(pts, fn.fused_params(), (H, W), n_points=n_points, channels_last=cl)
        torch.cuda.synchronize()
        assert canvas.shape == (3, 64, H, W)
        assert canvas.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        assert torch.equal(idx, idx2)
        assert torch.equal(canvas, want)
    assert (canvas[2] == 0).all()
    # stale contents of a reused canvas are cleared
    buf = torch.full((3, 64, H, W), 7.0, device=gpu).contiguous(memory_format=torch.channels_last)
    out_idx = torch.empty_like(idx)
    vox.pfn_canvas(pts, fn.fused_params(), (H, W), n_points=n_points, out=(buf, out_idx))
    assert torch.equal(buf, want)
    with pytest.raises(Exception):
        vox.pfn_canvas(pts, fn.fused_params(), (H - 1, W), n_points=n_points)
    # a canvas handed back call after call (pp_voxelize_pfn_canvas_reuse_dev): only the previous call's pixels
    # are cleared -- DIFFERENT clouds through the same buffers, each result equal to a fresh full-clear call
    for cl in (True, False):
        fmt = torch.channels_last if cl else torch.contiguous_format
        cbuf = torch.full((3, 64, H, W), 5.0, device=gpu).contiguous(memory_format=fmt)
        ibuf = torch.empty_like(idx)
        for k, seeds in enumerate([(1, 2, 3), (7, 8, 9), (4, 4, 4), (1, 2, 3)]):
            q = torch.from_numpy(np.stack([synth.lidar_like(20000, half, s_) for s_ in seeds])).to(gpu)
            npts = [20000, 700 + 3000 * k, 0 if k % 2 == 0 else 20000]
            fresh, fresh_idx = vox.pfn_canvas(q, fn.fused_params(), (H, W), n_points=npts, channels_last=cl)
            vox.pfn_canvas(q, fn.fused_params(), (H, W), n_points=npts, channels_last=cl, out=(cbuf, ibuf),
                           reuse=(k > 0))
            torch.cuda.synchronize()
            assert torch.equal(cbuf, fresh) and torch.equal(ibuf, fresh_idx), (cl, k)
    # PPScatter's own inference path (pp_scatter_canvas_dev) gives the same canvas, channels last
    sc2 = M.PPScatter(H, W).eval()
    with torch.no_grad():
        got = sc2(feats, idx)
        got24 = sc2(feats[:, :24].contiguous(), idx)          # a channel count that is not 64
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    assert torch.equal(got24, want[:, :24])
    # ... and through the C ABI into an NCHW canvas
    import ctypes
    from pp_amd import _lib
    nchw = torch.full((3, 64, H, W), 5.0, device=gpu)
    rc = _lib.lib().pp_scatter_canvas_dev(M._hip_ctx(gpu).handle, None, ctypes.c_void_p(feats.data_ptr()),
                                          ctypes.c_void_p(idx.data_ptr()), 3, 64, P, ctypes.c_void_p(nchw.data_ptr()),
                                          H, W, 0)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(nchw, want)


def test_fused_scatter_pipeline_equals_dense_pipeline(gpu):
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    pipe = PillarPipeline(VoxelConfig.square(16.0, 0.2, 4000, 32), feature_channels=64, device=gpu, seed=0)
    pipe.model.eval()
    pts = torch.from_numpy(np.stack([synth.lidar_like(15000, 16.0, s) for s in (3, 4)])).to(gpu)
    outs = {}
    for name, fs, cl in (("nhwc", True, True), ("feats", False, True), ("nchw", False, False)):
        pipe.fused_scatter = fs
        pipe.model.scatter.channels_last_inference = cl
        outs[name] = tuple(t.clone() for t in pipe.forward_fused(pts))
    outs["dense"] = pipe.forward(pts)
    torch.cuda.synchronize()
    c0, r0 = outs["nchw"]
    assert c0.shape[0] == 2 and c0.shape[1] == 18 and r0.shape[1] == 16
    for name in ("nhwc", "feats", "dense"):
        c, r = outs[name]
        assert c.shape == c0.shape and r.shape == r0.shape
        assert (c - c0).abs().max().item() <= 1e-4 * max(1.0, c0.abs().max().item()), name
        assert (r - r0).abs().max().item() <= 1e-4 * max(1.0, r0.abs().max().item()), name


def test_fused_scatter_toggled_between_clouds_never_sees_stale_pixels(gpu):
    """forward_fused clears only the pixels ITS previous fused-scatter call wrote.  Order that broke it: fused
    scatter on cloud A, the feature path (fused_scatter False) on cloud B, fused scatter on cloud C -- with one
    shared index buffer the sparse clear zeroed B's pixels and A's stayed.  Every call must equal a
    fresh pipeline's result for its cloud."""
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    cfg = VoxelConfig.square(16.0, 0.2, 4000, 32)
    pipe = PillarPipeline(cfg, feature_channels=64, device=gpu, seed=0)
    pipe.model.eval()
    clouds = [torch.from_numpy(np.stack([synth.lidar_like(n, r, s) for s in (sd, sd + 1)])).to(gpu)
              for n, r, sd in ((15000, 16.0, 3), (9000, 6.0, 40), (12000, 11.0, 80), (15000, 16.0, 3))]

    def fresh(pts):
        q = PillarPipeline(cfg, feature_channels=64, device=gpu, seed=0)
        q.model.eval()
        return tuple(t.clone() for t in q.forward_fused(pts))
    want = [fresh(c) for c in clouds]
    for k, (pts, fs) in enumerate(zip(clouds, (True, False, True, True))):
        pipe.fused_scatter = fs
        c, r = pipe.forward_fused(pts)
        torch.cuda.synchronize()
        tol = 1e-4 if not fs else 2e-6          # the same path up to MIOpen's choice of algorithm (a stale pixel
                                                # carries a whole feature vector: errors of 1e-2 and more)
        assert (c - want[k][0]).abs().max().item() <= tol * max(1.0, want[k][0].abs().max().item()), k
        assert (r - want[k][1]).abs().max().item() <= tol * max(1.0, want[k][1].abs().max().item()), k
    # a raised error must not leave a key that claims the canvas is known
    pipe.fused_scatter = True
    with pytest.raises(ValueError):
        pipe.forward_fused(clouds[0].double())
    assert pipe._canvas_key is None
    c, r = pipe.forward_fused(clouds[2])
    assert (c - want[2][0]).abs().max().item() <= 2e-6 and (r - want[2][1]).abs().max().item() <= 2e-6


def test_training_feature_net_matches_pytorch_autograd(gpu):
    """PPFeatureNet in train() on the HIP kernels (batch statistics, forward, parameter
    gradients; no [B,64,P,N] intermediate) against the PyTorch module sequence + autograd in
    f64 (model/model.py:31-40): output, running statistics, d/d(conv weight, conv bias,
    BN weight, BN bias).  Tolerance: f32 evaluation of sums over ~2M slots -> 2e-4 relative
    to each tensor's scale (PyTorch's own f32 path is checked to the same bar)."""
    import torch
    import pp_amd.model as M
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    vox = PillarVoxelizer(VoxelConfig.square(20.0, 0.2, 6000, 32), device=gpu)
    pts = torch.from_numpy(np.stack([synth.lidar_like(20000, 20.0, s) for s in (0, 1, 2)])).to(gpu)
    dense, _ = vox(pts)
    # keep magnitudes moderate so that f32 sums are meaningful: the reference subtracts a data mean
    dense = dense / torch.tensor([20, 20, 3, 255, 200, 200, 1, 1, 1], device=gpu).view(1, 9, 1, 1)
    results = {}
    for name, dtype, hip in (("f64", torch.float64, False), ("torch32", torch.float32, False),
                             ("hip", torch.float32, True)):
        torch.manual_seed(11)
        fn = M.PPFeatureNet(9, 64).to(gpu)
        with torch.no_grad():
            fn.bn1.weight.normal_(0, 1.0)      # both signs of the BatchNorm scale
            fn.bn1.bias.normal_(0, 0.3)
            fn.conv1.weight.mul_(2.0)
        fn = fn.to(dtype).train()
        fn.hip_train = hip
        x = dense.to(dtype)
        out = fn(x)
        gsrc = torch.Generator(device="cpu").manual_seed(5)
        g = torch.randn(out.shape, generator=gsrc).to(gpu).to(dtype)
        out.backward(g)
        results[name] = dict(out=out.detach().double(), rm=fn.bn1.running_mean.double(),
                             rv=fn.bn1.running_var.double(), nb=int(fn.bn1.num_batches_tracked),
                             dw=fn.conv1.weight.grad.double(), db=fn.conv1.bias.grad.double(),
                             dg=fn.bn1.weight.grad.double(), dbeta=fn.bn1.bias.grad.double())
    torch.cuda.synchronize()
    ref = results["f64"]
    assert (ref["dg"].abs() > 0).all() and results["hip"]["nb"] == 1

    def err(a, b):
        return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
    for key in ("out", "rm", "rv", "dw", "db", "dg", "dbeta"):
        e_hip, e_t32 = err(results["hip"][key], ref[key]), err(results["torch32"][key], ref[key])
        assert e_hip <= 2e-4, (key, e_hip, e_t32)
        assert e_hip <= 10 * e_t32 + 2e-5, (key, e_hip, e_t32)
    # a second step keeps accumulating the running statistics like BatchNorm2d does
    assert results["hip"]["rm"].abs().max() > 0


def test_training_pipeline_step_uses_hip_feature_net(gpu):
    """train_forward_backward with the HIP training feature net vs the same step on the
    PyTorch modules.  The forward agrees to f32 rounding (losses to 1e-5); the gradients of a
    deep f32 network with batch-statistics BatchNorm amplify that rounding (two PyTorch runs
    differ by up to 3e-2 of a tensor's scale from MIOpen's atomics alone), so they are compared
    by direction: cosine >= 0.999 per tensor group and overall."""
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import VoxelConfig
    cfg = VoxelConfig.square(16.0, 0.2, 3000, 32)
    pts = torch.from_numpy(np.stack([synth.lidar_like(12000, 16.0, s) for s in (0, 1)])).to(gpu)
    gts = [synth.gt_boxes(12, cfg.canvas_height, s) for s in (0, 1)]
    grads, losses = {}, {}
    for hip in (False, True):
        pipe = PillarPipeline(cfg, device=gpu, seed=0, with_targets=True)
        pipe.model.train()
        pipe.model.feature_net.hip_train = hip
        losses[hip] = [t.item() for t in pipe.train_forward_backward(pts, gts)]
        grads[hip] = {n: p.grad.detach().double().flatten() for n, p in pipe.model.named_parameters()}
    for a, b in zip(losses[True], losses[False]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (losses[True], losses[False])

    def cos(names):
        u = torch.cat([grads[True][n] for n in names])
        v = torch.cat([grads[False][n] for n in names])
        return (u @ v / (u.norm() * v.norm())).item()
    names = list(grads[False])
    assert cos(names) >= 0.999
    for prefix in ("feature_net.", "backbone.down1", "backbone.down2", "backbone.down3", "det_head."):
        assert cos([n for n in names if n.startswith(prefix)]) >= 0.999, prefix


def test_feature_net_kernels_odd_slot_counts(gpu):
    """N not a multiple of 4 (the LDS / scalar kernels instead of the MFMA one), P not a multiple
    of 4, a single sweep: eval output bit-identical to the fused voxelizer's, training output
    and gradients within tolerance of PyTorch's."""
    import torch
    import pp_amd.model as M
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    for N, P in ((7, 2001), (10, 1500), (301, 130)):
        vox = PillarVoxelizer(VoxelConfig.square(20.0, 0.5 if N > 100 else 0.2, P, N), device=gpu)
        fn = _net(gpu)
        pts = torch.from_numpy(synth.lidar_like(20000, 20.0, N)[None]).to(gpu)
        dense, idx = vox(pts)
        feats, idx2 = vox.pfn(pts, fn.fused_params())
        with torch.no_grad():
            hip = fn(dense)
            fn.hip_eval = False
            ref = fn(dense)
        torch.cuda.synchronize()
        assert torch.equal(hip, feats)
        assert (hip - ref).abs().max().item() <= TOL_ABS
        # training mode on the same tensor, against an f64 evaluation of the module sequence
        # (PyTorch's own f32 BatchNorm is off by percents here: most slots are zero padding, the
        # variance of a channel is tiny next to its mean; the HIP statistics are summed about the
        # padded value and do not cancel)
        x = dense / torch.tensor([20, 20, 3, 255, 200, 200, 1, 1, 1], device=gpu).view(1, 9, 1, 1)
        out = {}
        for name, dtype, hip_train in (("f64", torch.float64, False), ("hip", torch.float32, True)):
            torch.manual_seed(2)
            f2 = M.PPFeatureNet(9, 64).to(gpu).to(dtype).train()
            f2.hip_train = hip_train
            y = f2(x.to(dtype))
            y.square().mean().backward()
            out[name] = [t.double() for t in (y.detach(), f2.conv1.weight.grad, f2.conv1.bias.grad,
                                              f2.bn1.weight.grad, f2.bn1.bias.grad, f2.bn1.running_var)]
        for a, b in zip(out["hip"], out["f64"]):
            assert (a - b).abs().max().item() <= 2e-4 * max(1e-6, b.abs().max().item()), (N, P)


def test_fused_canvas_and_dense_feature_net_in_a_hip_graph(gpu):
    """pp_voxelize_pfn_canvas_dev (memset + four kernels) and pp_voxelize_dev + pp_pfn_dense_dev
    allocate and synchronise nothing once warm: both sequences capture into one HIP graph
    and replay on new inputs with the eager results."""
    import torch
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    P, N = 3000, 16
    vox = PillarVoxelizer(VoxelConfig.square(12.0, 0.2, P, N), device=gpu)
    H = W = vox.cfg.canvas_height
    fn = _net(gpu)
    params = fn.fused_params()
    clouds = [torch.from_numpy(synth.lidar_like(9000, 12.0, s)[None]).to(gpu) for s in (40, 41)]
    static_in = clouds[0].clone()
    canvas = torch.empty((1, 64, H, W), device=gpu).contiguous(memory_format=torch.channels_last)
    idx = torch.empty((1, P, 3), dtype=torch.int64, device=gpu)
    dense = (torch.empty((1, 9, P, N), device=gpu), torch.empty((1, P, 3), dtype=torch.int64, device=gpu))

    def step():
        vox.pfn_canvas(static_in, params, (H, W), out=(canvas, idx))
        vox(static_in, out=dense)
        with torch.no_grad():
            return fn(dense[0])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        feats = step()
    for c in (clouds[1], clouds[0], clouds[1]):
        static_in.copy_(c)
        g.replay()
        torch.cuda.synchronize()
        want_canvas, _ = vox.pfn_canvas(c, params, (H, W))
        want_feats, _ = vox.pfn(c, params)
        assert torch.equal(canvas, want_canvas)
        assert torch.equal(feats, want_feats)


def test_pipelined_fused_canvas_equals_three_launch_form_and_pytorch(gpu):
    """PillarVoxelizer.submit_pfn_canvas / pp_voxelize_step_pfn_canvas_dev (ONE launch per call: split | tile | order
    roles of three younger batches, the emit role as the fused feature net writing the canvas, and a CLEAR role for
    the other canvas): a sequence of DIFFERENT batches -- changing batch size, ragged counts, an empty sweep, an
    overflowing sweep -- comes out in order, each canvas bit-identical to pp_voxelize_pfn_canvas_reuse_dev's for
    that batch (both memory layouts), and equal to PyTorch's PPFeatureNet + PPScatter on the dense tensor
    (model/model.py:31-62) within f32 rounding.  Dense submits on the same pipeline in between stay exact."""
    import torch
    import pp_amd.model as M
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    half, step, P, N = 24.0, 0.2, 3000, 16
    cfg = VoxelConfig.square(half, step, P, N)
    H = W = cfg.canvas_height
    fn = _net(gpu)
    tab = fn.fused_params()
    ref_vox = PillarVoxelizer(cfg, device=gpu)
    seq = []
    for i, (B, n) in enumerate([(2, 9000), (2, 20000), (2, 5000), (3, 9000), (3, 12000), (1, 300), (1, 20000)]):
        pts = np.stack([synth.lidar_like(n, half, 50 + 7 * i + s) for s in range(B)])
        npts = [n - 37 * s for s in range(B)]
        if i == 2:
            npts[1] = 0
        seq.append((torch.from_numpy(pts).to(gpu), npts))
    for cl in (True, False):
        want = []
        for t, npts in seq:
            c, ix, cnt = ref_vox.pfn_canvas(t, tab, (H, W), n_points=npts, channels_last=cl, return_counts=True)
            want.append((c.clone(), ix.clone(), cnt.clone()))
        assert want[1][2][0, 0].item() > P                      # an overflowing sweep is in the sequence
        vs = PillarVoxelizer(cfg, device=gpu)
        for rep in range(2):                                     # twice: slots and canvases reused with stale contents
            got = []
            for k, (t, npts) in enumerate(seq):
                r = vs.submit_pfn_canvas(t, tab, (H, W), n_points=npts, channels_last=cl, return_counts=True)
                assert (r is None) == (k < vs.LAG)
                if r is not None:
                    got.append(tuple(x.clone() for x in r))      # valid until the next call only
            for _ in range(vs.LAG):
                got.append(tuple(x.clone() for x in vs.submit_pfn_canvas(None, tab, (H, W), channels_last=cl,
                                                                           return_counts=True)))
            assert vs.submit_pfn_canvas(None, tab, (H, W), channels_last=cl) is None
            torch.cuda.synchronize()
            assert len(got) == len(seq)
            for k, (w, g) in enumerate(zip(want, got)):
                assert g[0].is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
                assert torch.equal(w[1], g[1]) and torch.equal(w[2], g[2]), k
                assert torch.equal(w[0], g[0]), (cl, rep, k)
    # against PyTorch's modules on the dense tensor (the reference's own sequence of ops)
    sc = M.PPScatter(H, W)
    sc.channels_last_inference = False
    with torch.no_grad():
        for (t, npts), w in zip(seq[:3], want[:3]):
            pil, idx = ref_vox(t, n_points=npts)
            ref = sc(fn(pil), idx)
            assert (ref - w[0]).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # dense and fused calls share ONE pipeline: whatever call is made when a batch is due decides its form
    vs = PillarVoxelizer(cfg, device=gpu)
    outs = []
    for k, (t, npts) in enumerate(seq[:5] + [(None, None)] * vs.LAG):
        if k % 2 == 0:
            r = vs.submit_pfn_canvas(t, tab, (H, W), n_points=npts, channels_last=False)
        else:
            r = vs.submit(t, n_points=npts)
        outs.append(None if r is None else (k % 2, tuple(x.clone() for x in r)))
    torch.cuda.synchronize()
    got = [o for o in outs if o is not None]
    assert len(got) == 5
    for (kind, g), w, (t, npts) in zip(got, want, seq):
        if kind == 0:
            assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1])
        else:
            pil, idx = ref_vox(t, n_points=npts)
            assert torch.equal(g[0], pil) and torch.equal(g[1], idx)


def test_pipelined_fused_full_size_and_pipeline(gpu):
    """The pipelined fused form at BASELINE config 2's sizes (B = 4, 60k points, 500x500, P = 12000, N = 100),
    bit-identical to the three-launch fused call; and PillarPipeline.forward_fused_pipelined == forward_fused
    batch for batch."""
    import torch
    from pp_amd import synth
    from pp_amd.pipeline import PillarPipeline
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    from util import C2
    cfg = VoxelConfig.square(C2["half"], C2["step"], C2["P"], C2["N"])
    H = W = cfg.canvas_height
    tab = _net(gpu).fused_params()
    a, b = PillarVoxelizer(cfg, device=gpu), PillarVoxelizer(cfg, device=gpu)
    ts = [torch.from_numpy(np.stack([synth.lidar_like(C2["n"], C2["half"], 10 * i + s) for s in range(4)])).to(gpu)
          for i in range(3)]
    got = []
    for t in ts + [None] * b.LAG:
        r = b.submit_pfn_canvas(t, tab, (H, W))
        if r is not None:
            got.append((r[0].clone(), r[1].clone()))
    for t, g in zip(ts, got):
        c, ix = a.pfn_canvas(t, tab, (H, W))
        assert torch.equal(c, g[0]) and torch.equal(ix, g[1])
    pipe = PillarPipeline(VoxelConfig.square(16.0, 0.2, 4000, 32), feature_channels=64, device=gpu, seed=0)
    pipe.model.eval()
    clouds = [torch.from_numpy(np.stack([synth.lidar_like(n, 16.0, s) for s in (sd, sd + 1)])).to(gpu)
              for n, sd in ((15000, 3), (9000, 40), (12000, 80), (15000, 5))]
    want = [tuple(x.clone() for x in pipe.forward_fused(c)) for c in clouds]
    outs = [pipe.forward_fused_pipelined(c) for c in clouds] + \
           [pipe.forward_fused_pipelined(None) for _ in range(pipe.voxelizer.LAG)]
    outs = [o for o in outs if o is not None]
    torch.cuda.synchronize()
    assert len(outs) == len(clouds)
    for (c, r), (wc, wr) in zip(outs, want):
        assert (c - wc).abs().max().item() <= 2e-6 and (r - wr).abs().max().item() <= 2e-6


def test_pipelined_fused_and_dense_with_more_binning_blocks_than_the_chip_holds(gpu):
    """Twelve sweeps on a 500x500 grid: 12 x (123 tiles + chunks) + order blocks > the 1 280 workgroups the chip holds,
    so k_step spreads the binning blocks between the emit blocks instead of dispatching them first -- the fused form
    (with its clear role in front) and the dense form, both bit-identical to the three-launch calls."""
    import torch
    from pp_amd import synth
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    cfg = VoxelConfig.square(50.0, 0.2, 1500, 8)
    H = W = cfg.canvas_height
    tab = _net(gpu).fused_params()
    a, b, c = (PillarVoxelizer(cfg, device=gpu) for _ in range(3))
    ts = [torch.from_numpy(np.stack([synth.lidar_like(3000, 50.0, 100 * i + s) for s in range(12)])).to(gpu)
          for i in range(3)]
    got_f, got_d = [], []
    for t in ts + [None] * b.LAG:
        r = b.submit_pfn_canvas(t, tab, (H, W))
        if r is not None:
            got_f.append((r[0].clone(), r[1].clone()))
        d = c.submit(t)
        if d is not None:
            got_d.append((d[0].clone(), d[1].clone()))
    assert len(got_f) == len(got_d) == 3
    for t, f, d in zip(ts, got_f, got_d):
        cv, ix = a.pfn_canvas(t, tab, (H, W))
        assert torch.equal(cv, f[0]) and torch.equal(ix, f[1])
        pil, ix2 = a(t)
        assert torch.equal(pil, d[0]) and torch.equal(ix2, d[1])
