#!/usr/bin/env python3
"""Practical HBM ceilings of this GPU for DESIGN.md: write-only, read-only and copy streams of 16 GiB
(torch elementwise kernels, HIP-event timed)."""
import torch
n = 2 * 1024 ** 3   # doubles: 16 GiB
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
b = n * 8
print("write-only (fill_)   %.2f TB/s" % (b / t(lambda: x.fill_(1.5)) / 1e12))
print("read-only  (sum)     %.2f TB/s" % (b / t(lambda: x.sum()) / 1e12))
print("copy (read+write)    %.2f TB/s" % (2 * b / t(lambda: y.copy_(x)) / 1e12))
print("hipMemsetAsync-like zero_ %.2f TB/s" % (b / t(lambda: x.zero_()) / 1e12))
