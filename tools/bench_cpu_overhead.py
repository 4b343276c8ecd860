"""Dev tool: host-side cost per call of the custom autograd ops (tiny tensors: GPU time ~0)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd.model as M
dev = torch.device("cuda", 0)
bn = torch.nn.BatchNorm2d(8).to(dev).train()
z = torch.randn(2, 8, 16, 16, device=dev, requires_grad=True)
g = torch.randn(2, 8, 16, 16, device=dev)
cb = torch.zeros(8, device=dev, requires_grad=True)
def fused():
    y = M._relu_bn(z, bn, conv_bias=cb); y.backward(g)
def plain():
    y = M._relu_bn(z, bn, enabled=False, conv_bias=cb); y.backward(g)
for name, f in (("fused hip", fused), ("pytorch", plain)):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name}: {(t1 - t0) * 1e3:.1f} us host time per fwd+bwd call")
