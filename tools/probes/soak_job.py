#!/usr/bin/env python3
"""Soak: many sub-batches through the feeder / copier / sink threads - 4096 host-resident 256 x 456 frames (the reference's video
shape), three jobs back to back, results copied to the host and compared between jobs; then 512 frames written as PNG files."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n, h, w = int(os.environ.get("N", 4096)), 256, 456
wts = bench.synth_weights()
eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
style = torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev)
base = [synth.frame_u8_torch(7 + k, h, w, dev).cpu().numpy() for k in range(64)]
frames = [np.roll(base[k % 64], k // 64, axis=0) for k in range(n)]
out = [torch.empty((n, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
cache = {}
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, info = jobs.stylize_frames_sharded(eng, frames, style, style_cache=cache, host_out=out[rep % 2], gather=False, sink=lambda i, j, u8: None)
    dt = time.perf_counter() - t0
    print(f"job {rep}: {n} frames in {dt:.2f} s = {n / dt:.0f} frames/s ({n * h * w / 1e6 / dt:.0f} Mpixels/s); feeder {info['feeder']}", flush=True)
    if rep:
        assert torch.equal(out[0], out[1]), "jobs differ"
tmp = tempfile.mkdtemp(prefix="adain_soak_")
sink = jobs.FileSink(dev, workers=8)
t0 = time.perf_counter()
jobs.stylize_frames_sharded(eng, frames[:512], style, style_cache=cache, gather=False,
                            sink=lambda i, j, u8: sink.write(u8, [os.path.join(tmp, f"f{k:05d}.png") for k in range(i, j)]))
sink.close()
print(f"512 PNG files in {time.perf_counter() - t0:.2f} s; {len(os.listdir(tmp))} written", flush=True)
assert len(os.listdir(tmp)) == 512
print("soak ok")
