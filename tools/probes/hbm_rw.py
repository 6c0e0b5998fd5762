#!/usr/bin/env python3
"""Achievable HBM write / read / copy bandwidth with torch's own kernels on 268 MB (the size conv_first writes and conv_last reads)."""
import torch
n = 64 * 1024 * 1024 * 4 // 4
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mb = n * 4 / 1e6
for name, fn, b in (("fill (write)", lambda: x.fill_(1.0), mb), ("sum (read)", lambda: x.sum(), mb), ("copy (read+write)", lambda: y.copy_(x), 2 * mb)):
    us = t(fn)
    print(f"{name:18s} {us:7.1f} us  {b / us:.2f} TB/s")   # MB / us = TB/s
