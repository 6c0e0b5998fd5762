#!/usr/bin/env python3
"""How long do the first steps after a device synchronisation take?  bench.py times K = 20 steps straight after a barrier
(synchronize), the `sustained` leg times ~570 steps: this prints the per-step durations (HIP events) of 40 steps after a
synchronize, after idle gaps of 0 / 2 / 20 ms.   python tools/probes/step_ramp_probe.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
step = bench.Step(dev, config=2)
for _ in range(5):
    step.run()
torch.cuda.synchronize()
out = {}
for gap_ms in (0, 2, 20):
    rows = []
    for rep in range(3):
        for _ in range(10):
            step.run()
        torch.cuda.synchronize()
        time.sleep(gap_ms * 1e-3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        ev[0].record()
        for k in range(40):
            step.run()
            ev[k + 1].record()
        torch.cuda.synchronize()
        rows.append([ev[k].elapsed_time(ev[k + 1]) for k in range(40)])
    med = [sorted(r[k] for r in rows)[1] for k in range(40)]
    out[f"idle_{gap_ms}ms"] = {"first_8_steps_ms": [round(x, 3) for x in med[:8]], "mean_steps_9_40_ms": round(sum(med[8:]) / 32, 4),
                               "mean_first_20_ms": round(sum(med[:20]) / 20, 4)}
print(json.dumps(out), flush=True)
