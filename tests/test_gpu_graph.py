"""The device entry points launch kernels only (no allocation, no synchronisation, no host read-back
once their scratch exists): a warmed-up call can be captured in a HIP graph and replayed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _capture(fn):
    import torch
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()                                  # scratch, dynamic-LDS attributes, layout: first call
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = fn()
    torch.cuda.synchronize()
    return g, out


def test_voxelizer_targets_decoder_replay_from_a_graph(gpu, oracle):
    import torch
    from pp_amd import boxes, synth
    from pp_amd.postprocess import Detector
    from pp_amd.targets import TargetAssigner
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    vox = PillarVoxelizer(VoxelConfig.square(20.0, 0.2, 3000, 32), device=gpu)
    pts = torch.from_numpy(np.stack([synth.lidar_like(9000, 20.0, s) for s in range(2)])).to(gpu)
    ref = [t.clone() for t in vox(pts)]
    g, out = _capture(lambda: vox(pts))
    for _ in range(3):
        for t in out:
            t.fill_(3)
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, ref))
    vox.check()

    acfg = boxes.AnchorConfig(60, 60)
    ta = TargetAssigner(acfg, canvas_height=120, device=gpu)
    gt = synth.gt_boxes(12, 120, 3)
    gg = ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"])
    ref_t = [t.clone() for t in ta.assign_device(*gg)]
    g2, out_t = _capture(lambda: ta.assign_device(*gg))
    for _ in range(3):                        # the kernel's tail re-arms its scratch words every time
        for t in out_t:
            t.fill_(7)
        g2.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out_t, ref_t))
    assert (ref_t[1][:, 0] == 1).sum().item() > 0

    anchors = boxes.make_anchors(acfg)
    det = Detector(anchors, acfg, 120, 0.2, 0.2, -12.0, -12.0, device=gpu)
    rng = np.random.default_rng(5)
    cls = torch.from_numpy(rng.normal(-2.5, 1.5, (3, acfg.per_cell * 9, 60, 60)).astype(np.float32)).to(gpu)
    reg = torch.from_numpy(rng.normal(0, 0.3, (3, acfg.per_cell * 8, 60, 60)).astype(np.float32)).to(gpu)
    ref_d = [t.clone() for t in det(cls, reg)]
    g3, out_d = _capture(lambda: det(cls, reg))
    for _ in range(3):
        for t in out_d:
            t.zero_()
        g3.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out_d, ref_d))
    assert ref_d[2].min().item() > 0
