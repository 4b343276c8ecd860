"""Dev tool: per-layer conv forward+backward (input and weight gradients), NCHW vs NHWC,
MIOpen immediate mode (cudnn.benchmark False) as in training."""
import sys, time, torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LAYERS = [("down1.0 s2", "conv", 64, 64, 500, 2), ("down1.k s1", "conv", 64, 64, 250, 1),
          ("down2.0 s2", "conv", 64, 128, 250, 2), ("down2.k s1", "conv", 128, 128, 125, 1),
          ("down3.0 s2", "conv", 128, 256, 125, 2), ("down3.k s1", "conv", 256, 256, 63, 1),
          ("up1 s1", "convT", 64, 128, 250, 1), ("up2 s2", "convT", 128, 128, 125, 2),
          ("up3 s4", "convT", 256, 128, 63, 4), ("head 1x1", "conv1", 384, 102, 250, 1)]
COUNT = {"down1.k s1": 3, "down2.k s1": 5, "down3.k s1": 5}
def run(kind, x, w, s):
    if kind == "conv": return F.conv2d(x, w, None, s, 1)
    if kind == "conv1": return F.conv2d(x, w, None, 1, 0)
    return F.conv_transpose2d(x, w, None, s, 1, {1: 0, 2: 1, 4: 1}[s])
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot = {"nchw": 0.0, "nhwc": 0.0}
for name, kind, cin, cout, h, s in LAYERS:
    k = 1 if kind == "conv1" else 3
    res = {}
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
        w = (torch.randn((cin, cout, k, k) if kind == "convT" else (cout, cin, k, k), device="cuda") * 0.05)
        w = w.contiguous(memory_format=fmt).requires_grad_(True)
        y = run(kind, x, w, s)
        g = torch.randn_like(y)
        def step():
            x.grad = None; w.grad = None
            run(kind, x, w, s).backward(g)
        res[fmt_name] = timeit(step)
        tot[fmt_name] += COUNT.get(name, 1) * res[fmt_name]
    print(f"{name:12s} x{COUNT.get(name,1)} fwd+bwd nchw {res['nchw']:8.1f} us  nhwc {res['nhwc']:8.1f} us", flush=True)
print(f"B={B} benchmark={torch.backends.cudnn.benchmark}: sum nchw {tot['nchw']/1e3:.2f} ms, nhwc {tot['nhwc']/1e3:.2f} ms")
