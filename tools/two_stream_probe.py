#!/usr/bin/env python3
"""Does keeping two independent frames in flight on two HIP streams hide the per-kernel tails (every block of a
launch finishes and stores at the same moment)?  Compares frames/s of one stream vs two streams, config-4 frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = [bench.Step(dev, config=cfg, seed_offset=i) for i in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
N = 12


def one_stream():
    for i in range(N):
        steps[i & 1].run()


def two_streams():
    for i in range(N):
        with torch.cuda.stream(streams[i & 1]):
            steps[i & 1].run()


for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("one stream", one_stream), ("two streams", two_streams)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mp = N * steps[0].h * steps[0].w / 1e6 / dt
    print(f"config {cfg} {name:12s}: {dt / N * 1e3:8.3f} ms/frame  {mp:7.2f} Mpix/s  {steps[0].flops_per_step() * N / dt / 1e12:6.1f} TF/s")
