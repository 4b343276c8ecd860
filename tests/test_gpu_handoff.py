"""The target kernels' hand-off to a sample's last workgroup, A/B against the architecturally guaranteed form.

The shipped kernels (csrc/pp_iou.hip: k_targets, k_targets_gt) publish list entries and rows with write-through
stores, drain, and take a two-level agent-scope ticket; the last arriver reads with agent-scope loads and no acquire
fence -- a form MI355X_MICROARCH.md lists as measured, not guaranteed.  ``variants/
libpp_hip_strict.so`` is the same source built with -DPP_STRICT_HANDOFF: an agent-scope release fence in front of
every ticket, no early ticket, an agent-scope acquire + barrier at the head of the tail.  Both libraries are loaded
into this process and must give the same BITS on the same inputs -- over the fuzz cases' generators, and call by call
inside the alternating-input soak, where consecutive calls carry different boxes, so a stale read of the previous
call's list cannot return the right bytes.  (create_target, /root/reference utils/box_utils.py:193-228.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def strict_lib():
    import os
    from pp_amd import _lib
    path = _lib.variant_path("strict")
    if not os.path.exists(path):
        _lib.build_variant("strict")          # hipcc is on the GPU box's image too; normally built by build()
    return _lib.variant_lib("strict")


def _fuzz_case(rng):
    """One random anchor grid + one random batch (the generators of tests/test_gpu_fuzz.py, condensed)."""
    from pp_amd import boxes
    fm = int(rng.integers(24, 64))
    per_cell = int(rng.integers(1, 10))
    scale = float(rng.choice([0.5, 0.5, 0.25, 0.4]))
    H = int(round(fm / scale))
    unit = H / (2.0 * fm)
    dims = tuple(tuple(float(v) for v in (rng.uniform(4, 14) * unit, rng.uniform(8, 30) * unit, rng.uniform(1, 3)))
                 for _ in range(per_cell))
    yaws = tuple(float(rng.choice([0.0, 90.0, 30.0])) for _ in range(per_cell))
    zs = tuple(float(rng.uniform(0.3, 1.2)) for _ in range(per_cell))
    acfg = boxes.AnchorConfig(fm, fm, scale, dims, yaws, zs)
    classes = int(rng.choice([3, 9, 12, 70]))
    gts = []
    for b in range(int(rng.integers(1, 6))):
        G = int(rng.choice([0, 1, 5, 30, 64, 65, 150, 700]))
        c = np.column_stack([rng.uniform(-5, H + 5, G), rng.uniform(-5, H + 5, G), rng.uniform(0, 2, G)])
        if G >= 10:
            c[:G // 2, :2] = rng.uniform(0.2 * H, 0.8 * H, 2) + rng.uniform(-6, 6, (G // 2, 2))
        g = {"centers": c,
             "wlh": np.column_stack([rng.uniform(4, 14, G) * unit, rng.uniform(8, 30, G) * unit, rng.uniform(1, 3, G)]),
             "yaw": rng.uniform(-np.pi, np.pi, G), "classes": rng.integers(0, classes, G).astype(np.int32)}
        if G >= 4:
            for k in ("centers", "wlh", "yaw"):
                g[k][1] = g[k][0]
        gts.append(g)
    return acfg, H, classes, gts


def test_strict_build_is_a_different_code_object(gpu, strict_lib):
    from pp_amd import _lib
    assert strict_lib is not _lib.lib()
    assert strict_lib._name != _lib.lib()._name
    assert strict_lib.pp_version() == _lib.lib().pp_version()


@pytest.mark.parametrize("case", list(range(16)))
def test_fuzz_cases_bit_equal_in_both_builds(gpu, oracle, strict_lib, case):
    """Random grids (1-9 anchor types: box-centric and anchor-centric kernels), random ragged batches (0-700 boxes a
    sample: every tail), both anchor sources, the batch form and the single-sample form: shipped == strict, bit for
    bit, and sample 0 against the oracle."""
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from util import check_targets, oracle_targets_for
    rng = np.random.default_rng(31000 + case)
    acfg, H, classes, gts = _fuzz_case(rng)
    anchors = boxes.make_anchors(acfg)
    ref = oracle_targets_for(oracle, anchors, gts[0], H, 0.5, classes)
    for src in (acfg, anchors):
        a = TargetAssigner(src, canvas_height=H, pos_thresh=0.5, num_classes=classes, device=gpu)
        b = TargetAssigner(src, canvas_height=H, pos_thresh=0.5, num_classes=classes, device=gpu, lib_=strict_lib)
        for _ in range(2):
            ca, ra = a.assign_batch(gts, check=True)
            cb, rb = b.assign_batch(gts, check=True)
        torch.cuda.synchronize()
        assert torch.equal(ca, cb) and torch.equal(ra, rb), (case, "batch")
        check_targets(ca[0], ra[0], *ref)
        g = gts[-1]
        c1, r1 = a.assign(g["centers"], g["wlh"], g["yaw"], g["classes"], check=True)
        c2, r2 = b.assign(g["centers"], g["wlh"], g["yaw"], g["classes"], check=True)
        assert torch.equal(c1, c2) and torch.equal(r1, r2), (case, "single")
        assert torch.equal(c1, ca[-1]) and torch.equal(r1, ra[-1])


@pytest.mark.parametrize("shape", ["config3", "reference_default"])
def test_alternating_soak_bit_equal_in_both_builds(gpu, oracle, strict_lib, shape):
    """The alternating-input soak of tests/test_gpu_targets.py with the two libraries side by side: call k runs box
    set order[k] through the shipped library and through the strict one (each on its own context, the shipped one's
    launches back to back with traffic on a second stream); the two results of every call must be the same bits, and
    the first call of every set equals the oracle's targets.  Also: what the strict form costs per launch."""
    import time
    import torch
    from pp_amd import boxes
    from pp_amd.targets import TargetAssigner
    from util import SideTraffic, check_targets, oracle_targets_for, soak_gt_sets, soak_order
    cfg, H = (boxes.AnchorConfig(250, 250), 500) if shape == "config3" else (boxes.AnchorConfig.reference_default(), 600)
    anchors = boxes.make_anchors(cfg)
    sets = soak_gt_sets(H)
    refs = [oracle_targets_for(oracle, anchors, s, H, 0.6) for s in sets]
    order = soak_order(len(sets), 200)
    for src in (cfg, anchors):
        a = TargetAssigner(src, canvas_height=H, device=gpu)
        b = TargetAssigner(src, canvas_height=H, device=gpu, lib_=strict_lib)
        g = [a._gt_to_device(s["centers"], s["wlh"], s["yaw"], s["classes"]) for s in sets]
        bad = torch.zeros((), dtype=torch.int64, device=gpu)
        seen = {}
        traffic = SideTraffic(gpu)
        for it, k in enumerate(order):
            if it % 3 == 0:
                traffic.flush()
            if it % 2 == 0:
                traffic.poke()
            ca, ra = a.assign_device(*g[k])
            cb, rb = b.assign_device(*g[k])
            bad += (torch.ne(ca, cb).any() | torch.ne(ra, rb).any()).to(torch.int64)
            seen.setdefault(k, (ca, ra))
        traffic.close()
        torch.cuda.synchronize()
        for k, (c, r) in seen.items():
            check_targets(c, r, *refs[k])
        assert int(bad.item()) == 0, f"{int(bad.item())} of {len(order)} calls: shipped != strict"
        # the price of the guaranteed form (event pairs, 100 calls of set 0 each, alternating three times)
        t = {"shipped": [], "strict": []}
        for _ in range(3):
            for name, ta in (("shipped", a), ("strict", b)):
                for _ in range(10):
                    ta.assign_device(*g[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    ta.assign_device(*g[0])
                e1.record()
                torch.cuda.synchronize()
                t[name].append(e0.elapsed_time(e1) * 10.0)        # us per call
        print(f"\nhandoff A/B {shape} {'grid' if src is cfg else 'arrays'}: shipped {min(t['shipped']):.2f} us, "
              f"strict {min(t['strict']):.2f} us per call (best of 3 x 100, allocator included)")
