"""Voxelizer-only micro benchmark (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa: E402
from pp_amd import synth  # noqa: E402
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60000)
ap.add_argument("--half", type=float, default=50.0)
ap.add_argument("--P", type=int, default=12000)
ap.add_argument("--N", type=int, default=100)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--order", type=int, default=1)
ap.add_argument("--step", type=float, default=0.2)
ap.add_argument("--pipelined", action="store_true",
                help="two-deep mode (PillarVoxelizer.submit): binning of batch i+1 overlaps the emit of batch i")
ap.add_argument("--cold", type=int, default=0,
                help="MiB of unrelated traffic (a fill of another buffer) between two calls, like the network's "
                     "activations between two voxelizer calls of the end-to-end step: the kernels' durations "
                     "(event pairs) then include the cold start")
ap.add_argument("--lanes", type=int, default=0,
                help="experiment: K independent (context, stream) lanes, batch i entirely on lane i %% K, no joins")
ap.add_argument("--rotate", type=int, default=1,
                help="K output buffers used in turn (a real consumer does not re-write one buffer call after call: with "
                     "K = 1 a buffer smaller than the 256 MB Infinity Cache is absorbed by it)")
a = ap.parse_args()

cfg = VoxelConfig.square(a.half, a.step, a.P, a.N, order=a.order)
vox = PillarVoxelizer(cfg)
pts = torch.from_numpy(np.stack([synth.lidar_like(a.n, a.half, s) for s in range(a.batch)])).cuda()
out = (torch.empty((a.batch, 9, a.P, a.N), dtype=torch.float32, device="cuda"),
       torch.empty((a.batch, a.P, 3), dtype=torch.int64, device="cuda"))
outs = [out] + [(torch.empty_like(out[0]), torch.empty_like(out[1])) for _ in range(a.rotate - 1)]
turn = {"k": 0}


def next_out():
    turn["k"] += 1
    return outs[turn["k"] % len(outs)]


call = (lambda: vox.submit(pts, out=next_out())) if a.pipelined else (lambda: vox(pts, out=next_out()))
if a.lanes:
    lanes = [(PillarVoxelizer(cfg), torch.cuda.Stream(),
              (torch.empty_like(out[0]), torch.empty_like(out[1]))) for _ in range(a.lanes)]
    state = {"i": 0}
    import ctypes
    from pp_amd import _lib as L_

    def call():
        v, st, o = lanes[state["i"] % a.lanes]
        state["i"] += 1
        n_arr = (ctypes.c_int32 * a.batch)(*([a.n] * a.batch))
        rc = L_.lib().pp_voxelize_dev(v._ctx.handle, ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(pts.data_ptr()),
                                      a.n, n_arr, a.batch, ctypes.byref(v._prm), ctypes.c_void_p(o[0].data_ptr()),
                                      ctypes.c_void_p(o[1].data_ptr()), None)
        assert rc == 0
for _ in range(20):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):       # wall time: event timing off (the event pairs cost launch time)
    call()
t_issue = (time.perf_counter() - t0) / a.iters
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(f"host: {t_issue * 1e6:.1f} us per call to issue, {dt * 1e6:.1f} us per call until the device is done")
vox.set_timing(a.iters + 2)
junk = torch.empty(a.cold << 20, dtype=torch.uint8, device="cuda") if a.cold else None
for _ in range(a.iters):       # the kernels, in a pass of their own
    call()
    if junk is not None:
        junk.zero_()
torch.cuda.synchronize()
if a.pipelined:
    for _ in range(vox.LAG):
        vox.submit(None, out=out)
    torch.cuda.synchronize()
from pp_amd import _lib  # noqa: E402
ks = np.median(vox.read_kernel_ms(_lib.KERNEL_SPLIT) or [float("nan")]) * 1e3
kt = np.median(vox.read_kernel_ms(_lib.KERNEL_TILE) or [float("nan")]) * 1e3
ms = vox.read_emit_ms(a.iters + 2)
print(f"kernels (median us): k_split {ks:.1f}  k_tile {kt:.1f}  k_emit {np.median(ms) * 1e3:.1f}"
      + ("  [pipelined: ONE kernel, k_step, in the k_emit column]" if a.pipelined else ""))
byt = cfg.algorithmic_bytes(a.n) * a.batch
emit = float(np.median(ms)) * 1e-3
print(f"batch={a.batch} n={a.n} P={a.P} N={a.N}: {dt*1e6:.1f} us/step  "
      f"{a.batch/dt:.0f} sweeps/s  emit median {emit*1e6:.1f} us "
      f"-> {byt/emit/1e12:.2f} TB/s algorithmic ({byt/emit/8e12*100:.1f}% of 8 TB/s); "
      f"whole pipeline {byt/dt/1e12:.2f} TB/s")
