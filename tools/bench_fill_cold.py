"""Calibration: zero-fill of the dense output (torch zero_) warm (same buffer back to back: the Infinity
Cache still holds it) and cold (1 GiB of other traffic in between, as the network's convolutions put
between two voxelizer calls).  The cold figure is what a store-only kernel can reach in the pipeline."""
import sys, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.empty((B, 9, 12000, 100), dtype=torch.float32, device="cuda")
junk = torch.empty((256 << 20,), dtype=torch.float32, device="cuda")   # 1 GiB
junk2 = torch.empty_like(junk)
def timed(n, between):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        between()
        a.record(); x.zero_(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
for _ in range(5): x.zero_()
warm = timed(50, lambda: None)
cold = timed(30, lambda: junk2.copy_(junk))
nb = x.numel() * 4
print(f"zero_ of {nb/1e6:.1f} MB: warm {warm*1e3:.1f} us = {nb/warm/1e9:.2f} TB/s; cold {cold*1e3:.1f} us = {nb/cold/1e9:.2f} TB/s")
