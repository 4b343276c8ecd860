"""Dev tool: per-layer f32 convolution timings of the backbone (model/model.py:112-160
shapes at the 500x500 canvas) in NCHW and NHWC, to see what MIOpen picks and costs."""
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda"

# (name, kind, cin, cout, hin, stride, extra)
LAYERS = [
    ("down1.0 s2", "conv", 64, 64, 500, 2), ("down1.k s1", "conv", 64, 64, 250, 1),
    ("down2.0 s2", "conv", 64, 128, 250, 2), ("down2.k s1", "conv", 128, 128, 125, 1),
    ("down3.0 s2", "conv", 128, 256, 125, 2), ("down3.k s1", "conv", 256, 256, 63, 1),
    ("up1 s1", "convT", 64, 128, 250, 1), ("up2 s2", "convT", 128, 128, 125, 2),
    ("up3 s4", "convT", 256, 128, 63, 4), ("head 1x1", "conv1", 384, 102, 250, 1),
]
COUNT = {"down1.k s1": 3, "down2.k s1": 5, "down3.k s1": 5}


def run(kind, x, w, stride):
    if kind == "conv":
        return F.conv2d(x, w, None, stride, 1)
    if kind == "conv1":
        return F.conv2d(x, w, None, 1, 0)
    op = {1: 0, 2: 1, 4: 1}[stride]
    return F.conv_transpose2d(x, w, None, stride, 1, op)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = {"nchw": 0.0, "nhwc": 0.0}
for name, kind, cin, cout, h, s in LAYERS:
    k = 1 if kind == "conv1" else 3
    x = torch.randn(B, cin, h, h, device=dev)
    w = torch.randn((cin, cout, k, k) if kind == "convT" else (cout, cin, k, k), device=dev) * 0.05
    y = run(kind, x, w, s)
    ho = y.shape[-1]
    if kind == "convT":
        flop = 2.0 * B * h * h * cin * cout * 9
    else:
        flop = 2.0 * B * ho * ho * cin * cout * k * k
    t0 = time.time()
    t_nchw = timeit(lambda: run(kind, x, w, s))
    xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    t_nhwc = timeit(lambda: run(kind, xl, wl, s))
    c = COUNT.get(name, 1)
    tot["nchw"] += c * t_nchw
    tot["nhwc"] += c * t_nhwc
    print(f"{name:12s} x{c} out {ho:3d}  {flop / 1e9:6.2f} GF  nchw {t_nchw:8.1f} us ({flop / t_nchw / 1e6:6.1f} TF/s)"
          f"  nhwc {t_nhwc:8.1f} us ({flop / t_nhwc / 1e6:6.1f} TF/s)  [{time.time() - t0:.0f}s]", flush=True)
print(f"B={B}: sum nchw {tot['nchw'] / 1e3:.2f} ms, nhwc {tot['nhwc'] / 1e3:.2f} ms, best-of {0:.0f}")
