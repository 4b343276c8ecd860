"""tools/lab/emit_stamps.py [B] [pfn|dense]: phase stamps of the emit waves of one three-launch call (library built with
-DPP_STAMPS=3: `build_full.sh estamps -DPP_STAMPS=3`).  Stamps per wave: 0 start, 2 descriptors in LDS + bucket read
issued, 3 indices stored / zero decisions done, 4 points staged + means + features in LDS, 5 folded / slab stored, 6 end."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd
from pp_amd import synth, _lib
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
what = sys.argv[2] if len(sys.argv) > 2 else "pfn"
step = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2      # 1.0: BASELINE configs[0]'s 100x100 grid (crowded cells)
P = 12000
pipe = PillarPipeline(VoxelConfig.square(50.0, step, P, 100), seed=0)
pipe.model.eval()
pts = torch.from_numpy(np.stack([synth.lidar_like(60000, 50.0, s) for s in range(B)])).cuda()
tab = pipe.model.feature_net.fused_table(pipe.device)
H, W = pipe.model.scatter.h, pipe.model.scatter.w
out = (pipe._canvas(B, H, W), torch.empty((B, P, 3), dtype=torch.int64, device="cuda"))
vox = pipe.voxelizer
dense = pipe._buffers(B)
for _ in range(20):
    if what == "pfn":
        vox.pfn_canvas(pts, tab, (H, W), out=out, reuse=True)
    else:
        vox(pts, out=dense)
torch.cuda.synchronize()
nbx = (P + 15) // 16
n = B * nbx * 16 * 8
buf = np.zeros(n, np.uint64)
f = _lib.lib().pp_debug_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
got = f(vox._ctx.handle, buf.ctypes.data, n)
st = buf[:got].reshape(-1, 16, 8)[:, :4, :].reshape(-1, 8).astype(np.int64)   # 4 waves per workgroup
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = lambda c: (c - t0) / 100.0
print(f"{what} B={B}: {len(st)} waves; us relative to the first start")
names = {0: "start", 2: "descriptors", 3: "indices/zeros", 4: "points staged", 5: "folded/stored", 6: "end"}
for k, nm in names.items():
    m = st[:, k] >= t0
    if m.any():
        c = us(st[m, k])
        print(f"  {nm:14s} n={m.sum():6d}  min {c.min():6.2f}  p10 {np.percentile(c,10):6.2f}  median {np.median(c):6.2f}  p90 {np.percentile(c,90):6.2f}  max {c.max():6.2f}")
life = (st[st[:, 6] >= t0, 6] - st[st[:, 6] >= t0, 0]) / 100.0      # (waves of rows beyond the pillar count leave early, unstamped)
print("  wave lifetime percentiles (us): p50 %.2f p90 %.2f p99 %.2f max %.2f; waves beyond 2 x median: %d" % (
    np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99), life.max(), (life > 2 * np.median(life)).sum()))
for a_, b_ in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (0, 6)):
    m = (st[:, a_] >= t0) & (st[:, b_] >= st[:, a_])
    if m.any():
        d = (st[m, b_] - st[m, a_]) / 100.0
        print(f"  {names[a_]:>14s} -> {names[b_]:14s} n={m.sum():6d} median {np.median(d):5.2f}  p90 {np.percentile(d,90):5.2f}  max {d.max():5.2f}")

# the slowest waves, stamp by stamp (non-pooled waves: stamp 4 = running means of all pillars done)
order = np.argsort(-(st[:, 6] - st[:, 0]))[:12]
print("  slowest waves (us from the wave's own start): descriptors, indices/zeros, stamp 4, stored, end")
for i in order:
    r = st[i]
    rel = lambda k: (r[k] - r[0]) / 100.0 if r[k] >= r[0] else float("nan")
    print("    start %6.2f | %5.2f %5.2f %5.2f %5.2f %5.2f" % (us(r[0]), rel(2), rel(3), rel(4), rel(5), rel(6)))

try:
    g = _lib.lib().pp_debug_means_prof
    g.argtypes = [ctypes.c_void_p, ctypes.c_int]
    prof = np.zeros(8, np.uint64)
    if g(prof.ctypes.data, 1) == 0 and prof[0]:
        print("  streamed_means, slowest wave (shader clocks): total %d, chains %d (%d steps in %d rounds: %.1f clocks per step), "
              "operand staging %d, next round's plan + fetch issue %d" % (prof[0], prof[1], prof[3], prof[4], prof[1] / max(1, prof[3]), prof[2], prof[5]))
except AttributeError:
    pass
