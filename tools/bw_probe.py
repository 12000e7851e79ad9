#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box: device-to-device copy / fill / read-only reduction of a 2.8 GB tensor (the size of
Conv2d_2b's output), as the yardstick for the HBM-bound kernels' fractions (roofline_hbm in the bench line is quoted against the 8 TB/s spec)."""
import torch, time
x = torch.empty(96 * 357 * 637 * 64, dtype=torch.bfloat16, device="cuda").normal_()
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 2 / 1e9
print(f"tensor {gb:.2f} GB")
dt = t(lambda: y.copy_(x)); print(f"copy  (read + write)  {2 * gb / dt / 1e3:.2f} TB/s  ({dt * 1e6:.0f} us)")
dt = t(lambda: y.zero_()); print(f"fill  (write only)    {gb / dt / 1e3:.2f} TB/s  ({dt * 1e6:.0f} us)")
dt = t(lambda: x.view(torch.int16).max()); print(f"max   (read only)     {gb / dt / 1e3:.2f} TB/s  ({dt * 1e6:.0f} us)")
xf = x.view(torch.float32); yf = y.view(torch.float32)
dt = t(lambda: torch.add(xf, 1.0, out=yf)); print(f"add f32 (read + write) {2 * gb / dt / 1e3:.2f} TB/s  ({dt * 1e6:.0f} us)")
