#!/usr/bin/env python3
"""Eager vs hipGraph replay of engine.stylize at the reference's default sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.engine import AdaINEngine, GraphedStylize

dev = torch.device("cuda", 0)
e = AdaINEngine(synth.to_torch(synth.vgg_state_dict(0, full=False)), synth.to_torch(synth.decoder_state_dict(0)), dev)
e.set_style(torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev))
for n, size in ((1, 256), (1, 512), (4, 256), (1, 1024)):
    x = torch.from_numpy(synth.image(3, n, size, size)).to(dev)
    g = GraphedStylize(e, n, size, size)
    ref = e.stylize(x).clone()
    out = g(x)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "graph replay must be bitwise identical to the eager path"
    res = []
    for name, fn in (("eager", lambda: e.stylize(x)), ("graph", lambda: g(x))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        res.append(f"{name} {dt * 1e3:7.3f} ms ({n * size * size / 1e6 / dt:7.1f} Mpix/s)")
    print(f"{n} x {size}x{size}: " + " | ".join(res))
