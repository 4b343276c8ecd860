"""Dev tool: PPFeatureNet training forward+backward alone (PyTorch path) on a config-2 dense tensor."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd.model as M
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(B, 9, 12000, 100, device="cuda")
fn = M.PPFeatureNet(9, 64).cuda().train()
g = torch.randn(B, 64, 12000, device="cuda")
def run():
    fn.zero_grad(set_to_none=True)
    y = fn(x)
    y.backward(g)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
print(f"PPFeatureNet train fwd+bwd B={B}: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
