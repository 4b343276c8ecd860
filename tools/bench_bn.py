"""Dev tool: BatchNorm2d training forward+backward, MIOpen vs PyTorch's native kernels."""
import torch, time
import torch.nn.functional as F
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for shape in ((4, 64, 250, 250), (4, 128, 125, 125), (4, 256, 63, 63), (4, 128, 250, 250), (4, 64, 12000, 100)):
    x = torch.randn(shape, device="cuda", requires_grad=True)
    C = shape[1]
    w = torch.ones(C, device="cuda", requires_grad=True); bb = torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    g = torch.randn(shape, device="cuda")
    def run():
        y = F.batch_norm(F.relu(x), rm, rv, w, bb, True, 0.1, 1e-5)
        y.backward(g)
    us_m = t(run)
    def run2():
        with torch.backends.cudnn.flags(enabled=False):
            y = F.batch_norm(F.relu(x), rm, rv, w, bb, True, 0.1, 1e-5)
        y.backward(g)
    us_n = t(run2)
    print(f"{shape}: relu+bn fwd+bwd  MIOpen {us_m:.0f} us   native {us_n:.0f} us   ({x.numel()*4/1e6:.0f} MB)")
