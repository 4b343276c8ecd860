"""Target-assignment micro benchmark (development aid; tools/collect_profiles.sh runs it under rocprofv3):
BASELINE config 3 by default.
usage: bench_targets.py [fm] [G] [batch] [per_cell: 2 | 6 (the reference's shipped anchor set)] [what: all | batch | single | grid (both, anchors on the fly only)]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import boxes, synth
from pp_amd.targets import TargetAssigner
fm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
six = len(sys.argv) > 4 and sys.argv[4] == "6"
what = sys.argv[5] if len(sys.argv) > 5 else "all"
ref = boxes.AnchorConfig.reference_default()
cfg = boxes.AnchorConfig(fm, fm, 0.5, ref.dims, ref.yaws_deg, ref.zs) if six else boxes.AnchorConfig(fm, fm)
gts = [synth.gt_boxes(G, 2 * fm, s) for s in range(B)]
it = 200
for name, src in (("arrays", boxes.make_anchors(cfg)), ("grid", cfg)):
    if what != "all" and name == "arrays":
        continue
    ta = TargetAssigner(src, canvas_height=2 * fm)
    A = ta.A
    if what in ("all", "single", "grid"):
        g = [ta._gt_to_device(gt["centers"], gt["wlh"], gt["yaw"], gt["classes"]) for gt in gts]
        for _ in range(10):
            ta.assign_device(*g[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(it):
            ta.assign_device(*g[k % B])
        t_issue = (time.perf_counter() - t0) / it       # the host's side alone: two allocations + the C call + the launch
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        print(f"A={A} G={G} anchors={name}: one sample per launch {dt*1e6:.1f} us per sample; algorithmic 112*A = "
              f"{112*A/1e6:.1f} MB -> {112*A/dt/1e9:.0f} GB/s")
        # the same loop three more times, apart: wall, the host's issue time, and the device's own span (event pair);
        # then with the outputs re-used (no allocator calls).  A loop whose wall time equals its issue time is bound by
        # the HOST (boxes of this pool differ 4x in single-thread speed), whatever the kernel takes.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for reuse in (False, False, False, True):
            out = (torch.empty((A, 9), dtype=torch.float32, device="cuda"), torch.empty((A, 9), dtype=torch.float32, device="cuda")) if reuse else None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for k in range(it):
                ta.assign_device(*g[k % B], out=out)
            e1.record()
            ti = (time.perf_counter() - t0) / it
            torch.cuda.synchronize()
            tw = (time.perf_counter() - t0) / it
            print(f"    loop of {it}{' (outputs re-used)' if reuse else ''}: wall {tw*1e6:.1f} us per call, host issue {ti*1e6:.1f}, "
                  f"device span (event pair) {e0.elapsed_time(e1)*1e3/it:.1f}")
    if what in ("all", "batch", "grid"):
        counts, packed = ta.upload_batch(gts)
        out = None
        for _ in range(10):
            out = ta.assign_batch_device(counts, packed, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            ta.assign_batch_device(counts, packed, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        print(f"A={A} G={G} anchors={name}: batch of {B} per launch {dt*1e6:.1f} us per call = {dt*1e6/B:.2f} us per sample; "
              f"112*A*B = {112*A*B/1e6:.1f} MB -> {112*A*B/dt/1e9:.0f} GB/s = {112*A*B/dt/8e12:.3f} of 8 TB/s")
