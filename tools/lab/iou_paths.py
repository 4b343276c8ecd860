"""tools/lab/iou_paths.py: what the cheap paths of k_targets cost -- the batch of 4 at BASELINE config 3 with
(a) the normal boxes, (b) every box moved far off the canvas (no workgroup has a near box: prologue + zero rows + ticket
only), (c) boxes with centres in range but tiny (pairs pass the gate, IoU ~ 0: gate + clip, no positives)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd
from pp_amd import boxes, synth
from pp_amd.targets import TargetAssigner
fm, G, B = 250, 40, 4
if len(sys.argv) > 1:
    fm = int(sys.argv[1])
ref = boxes.AnchorConfig.reference_default()
cfg = boxes.AnchorConfig(fm, fm) if fm == 250 else boxes.AnchorConfig(fm, fm, 0.5, ref.dims, ref.yaws_deg, ref.zs)
ta = TargetAssigner(cfg, canvas_height=2 * fm)
def run(name, gts):
    counts, packed = ta.upload_batch(gts)
    out = None
    for _ in range(10):
        out = ta.assign_batch_device(counts, packed, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ta.assign_batch_device(counts, packed, out=out)
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per call (A={ta.A}, B={len(gts)})")
gts = [synth.gt_boxes(G, 2 * fm, s) for s in range(B)]
run("normal", gts)
far = [dict(g, centers=g["centers"] + np.array([1e5, 1e5, 0])) for g in gts]
run("all boxes far away", far)
run("no boxes (G=0)", [{k: v[:0] for k, v in g.items()} for g in gts])
run("one box per sample", [{k: v[:1] for k, v in g.items()} for g in gts])
run("normal, B=1", gts[:1])
run("far, B=1", far[:1])
