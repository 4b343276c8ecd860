"""Host drop-in timing (PCIe-inclusive): pillars.create_pillars / make_ious on numpy arrays,
next to the CPU oracle's reference-style implementation.  Development aid; lives under tests/ because it times the CPU oracle beside the product (only tests/, smoke() and bench.py's cpu_baseline may touch oracle/)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd
from pp_amd import pillars, synth, boxes
from oracle import oracle as O

def med(fn, n=15):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

pts = synth.lidar_like(60000, 50.0, 0).astype(np.float64)
P, N = 12000, 100
args = (N, P, .2, .2, -50, -50, -10, 50, 50, 10, 500)
T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
def gpu_call():
    T[:] = 0; I[:] = 0          # caller zeroes (dataset.py:89-90); counted, like np.zeros is
    pillars.create_pillars(pts, T, I, *args)
def cpu_call():
    T[:] = 0; I[:] = 0
    O.create_pillars(pts, T, I, *args, order=O.ORDER_HASH)
print(f"create_pillars host drop-in (C2 shapes, incl. re-zeroing the 86 MB f64 tensor): HIP {med(gpu_call):.1f} ms, CPU oracle (hash) {med(cpu_call):.1f} ms")
def gpu_only():
    pillars.create_pillars(pts, T, I, *args)
def cpu_only():
    O.create_pillars(pts, T, I, *args, order=O.ORDER_HASH)
print(f"create_pillars call alone: HIP {med(gpu_only):.2f} ms, CPU oracle (hash) {med(cpu_only):.1f} ms")
anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
gt = synth.gt_boxes(40, 500, 0)
c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)
ious = np.zeros((125000, 40))
print(f"make_ious host drop-in (A=125000, G=40): HIP {med(lambda: pillars.make_ious(anchors['corners'], k_img, anchors['centers'], c_img, ious), 8):.1f} ms, "
      f"CPU oracle {med(lambda: O.make_ious(anchors['corners'], k_img, anchors['centers'], c_img, ious), 5):.1f} ms")
