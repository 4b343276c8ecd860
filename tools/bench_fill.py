"""Calibration: how fast can this chip zero-fill the dense output (torch zero_)?"""
import sys, time, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.empty((B, 9, 12000, 100), dtype=torch.float32, device="cuda")
for _ in range(20): x.zero_()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
for a, b in ev:
    a.record(); x.zero_(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[50]
print(f"zero_ B={B}: {ms*1e3:.1f} us -> {x.numel()*4/ms/1e9:.2f} TB/s")
