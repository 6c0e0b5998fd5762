#!/usr/bin/env python3
"""Achievable HBM write / read / copy bandwidth with torch's own kernels: on ONE 268 MB buffer (the size conv_first writes and conv_last
reads at 1024 x 1024; a single buffer of that size largely lives in the 256 MB Infinity Cache between repeats) and ROTATING over eight
such buffers (2.1 GB: every pass goes to HBM - the yardstick for a layer whose 268 MB are written once and read once)."""
import torch

n = 64 * 1024 * 1024 * 4 // 4
bufs = [torch.empty(n, device="cuda") for _ in range(8)]
outs = [torch.empty(n, device="cuda") for _ in range(2)]


def t(fn, reps=24):
    for k in range(8):
        fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


mb = n * 4 / 1e6
for name, fn, b in (("fill (write), one buffer", lambda k: bufs[0].fill_(1.0), mb),
                    ("fill (write), rotating 8", lambda k: bufs[k % 8].fill_(1.0), mb),
                    ("sum (read), one buffer", lambda k: bufs[0].sum(), mb),
                    ("sum (read), rotating 8", lambda k: bufs[k % 8].sum(), mb),
                    ("copy (read+write), one pair", lambda k: outs[0].copy_(bufs[0]), 2 * mb),
                    ("copy (read+write), rotating", lambda k: outs[k % 2].copy_(bufs[k % 8]), 2 * mb)):
    us = t(fn)
    print(f"{name:30s} {us:7.1f} us  {b / us:.2f} TB/s  ({268.4 / us * (2 if 'copy' in name else 1) / 8.0:.2f} of 8 TB/s)")   # MB / us = TB/s
