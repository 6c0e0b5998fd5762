#!/usr/bin/env python3
"""Where a host-resident job loses against an HBM-resident one: the same 128-frame 1080p job with frames resident / on the host,
with and without the result copied back, sub-batches of 2 / 4, feeder depth 3 / 6.  Prints one line per variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n, h, w = int(os.environ.get("N", 128)), 1080, 1920
wts = bench.synth_weights()
eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
style = torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev)
cache = {}
res = bench.FrameStore(4, n, 0, n, h, w, dev, host=False)
host = bench.FrameStore(4, n, 0, n, h, w, dev, host=True)
out = torch.empty((n, h, w, 3), dtype=torch.uint8).pin_memory()


def run(name, frames, sub, host_out=None, prefetch=3, reps=3):
    ts = []
    for r in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, info = jobs.stylize_frames_sharded(eng, frames, style, sub_batch=sub, style_cache=cache, out_hw=(h, w), host_out=host_out, prefetch=prefetch)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:44s} best {min(ts[1:])*1e3:8.1f} ms  all {[round(t*1e3) for t in ts[1:]]}  compute_s {info['compute_s']:.3f} feeder {info['feeder']}", flush=True)


for sub in (2, 4):
    run(f"resident sub={sub}", res, sub)
    run(f"resident sub={sub} + D2H", res, sub, host_out=out)
    run(f"host sub={sub} no D2H", host, sub)
    run(f"host sub={sub} + D2H", host, sub, host_out=out)
    run(f"host sub={sub} + D2H, prefetch 6", host, sub, host_out=out, prefetch=6)
