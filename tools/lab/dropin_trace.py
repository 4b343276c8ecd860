"""tools/lab/dropin_trace.py: where a call of the pybind11 drop-in module goes (PP_DROPIN_TRACE laps on stderr), for a
few pool sizes and chunk counts; then bench.py's dropin_host record.  Development aid (GPU box)."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import importlib.util, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import pp_amd
from pp_amd import _lib, synth, boxes
spec = importlib.util.spec_from_file_location("pillars", _lib.pybind_module_path())
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
pts = synth.lidar_like(60000, 50.0, 0).astype(np.float64)
agg = np.ascontiguousarray(pts.T)
P, N = 12000, 100
args = (N, P, .2, .2, -50., -50., -10., 50., 50., 10., 500)
T, I = np.zeros((P, N, 9)), np.zeros((P, 3))
def med(fn, n):
    for _ in range(4): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, np.min(ts) * 1e3
m, lo = med(lambda: mod.create_pillars(agg.transpose([1, 0]), T, I, *args), 40)
print(f"create_pillars call: median {m:.3f} ms, min {lo:.3f} ms", flush=True)
anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
gt = synth.gt_boxes(40, 500, 0)
c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)
ious = np.zeros((125000, 40))
m, lo = med(lambda: mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious), 20)
print(f"make_ious call: median {m:.3f} ms, min {lo:.3f} ms", flush=True)
"""

COMBOS = [  # (threads, h2d parts, chunks, send-ahead, polled waits)
    (1, 1, 1, 0, 0), (8, 1, 1, 0, 0), (8, 1, 1, 0, 1), (8, 2, 1, 0, 1), (8, 4, 1, 0, 1), (8, 2, 2, 0, 1),
    (8, 2, 1, 1, 1), (8, 2, 2, 1, 1), (8, 2, 4, 1, 1), (8, 1, 2, 1, 1), (4, 2, 2, 1, 1), (16, 2, 2, 1, 1)]
if len(sys.argv) > 1:
    COMBOS = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for threads, parts, chunks, spec, spin in COMBOS:
    env = dict(os.environ, PP_HOST_THREADS=str(threads), PP_DROPIN_H2D_PARTS=str(parts), PP_DROPIN_CHUNKS=str(chunks),
               PP_DROPIN_SPEC=str(spec), PP_DROPIN_SPIN=str(spin))
    print(f"== threads={threads} h2d_parts={parts} chunks={chunks} send_ahead={spec} polled_waits={spin}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, check=False)
    env["PP_DROPIN_TRACE"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
    tr = [l for l in r.stderr.splitlines() if l.startswith("pp_")]
    for name in ("pp_create_pillars_f64", "pp_make_ious_f64"):
        mine = [l for l in tr if l.startswith(name)]
        if mine:
            print("   trace (last call):", mine[-1], flush=True)
