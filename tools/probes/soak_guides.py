#!/usr/bin/env python3
"""Soak of round 5's threaded device transform: 600 PIL views of mixed sizes through `precompute_guides_sharded` (the views are fetched,
uploaded and resized ON THE DEVICE from the feeder's pool threads: per-thread pinned staging, one lock around a resize's two launches),
twice, files compared byte for byte between the runs and - for a sample - against the same call with the host (PIL) transform."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image

import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.AdaIN import test as T

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = int(os.environ.get("N", 600))
wts = bench.synth_weights()
eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
style = torch.from_numpy(synth.image(4, 1, 256, 256)).to(dev)
sizes = [(400, 400), (300, 520), (360, 480), (256, 256)]
base = {s: [(synth.image(100 + k, 1, s[0], s[1])[0].transpose(1, 2, 0) * 255).astype(np.uint8) for k in range(8)] for s in sizes}
views = [Image.fromarray(np.roll(base[sizes[(k // 25) % 4]][k % 8], k % 17, axis=1)) for k in range(n)]     # runs of 25 equal-sized views
masks = [np.asarray(v).transpose(2, 0, 1) > 40 for v in views]
names = [f"view_{k:04d}" for k in range(n)]
roots = [tempfile.mkdtemp(prefix="adain_soak_guides_") for _ in range(2)]
for rep, root in enumerate(roots):
    t0 = time.perf_counter()
    paths, info = jobs.precompute_guides_sharded(eng, views, names, root, style, masks=masks, content_size=256, save_ext=".png", write="local", writers=8)
    dt = time.perf_counter() - t0
    print(f"run {rep}: {n} views in {dt:.2f} s = {n / dt:.0f} views/s; files {len(os.listdir(root))}", flush=True)
a, b = roots
bad = [nm for nm in names if open(os.path.join(a, nm + ".png"), "rb").read() != open(os.path.join(b, nm + ".png"), "rb").read()]
assert not bad, bad[:5]
# a sample against the host transform: the same view resized by PIL, then the same engine call
tf = T.test_transform_u8(256, False)
for k in range(0, n, 37):
    host = tf(views[k])
    m = torch.from_numpy(masks[k][None]).to(dev)
    want = eng.stylize_u8(torch.from_numpy(host[None].copy()).to(dev), alpha=0.5, masks=m)[0].cpu().numpy()
    got = np.asarray(Image.open(os.path.join(a, names[k] + ".png")))
    assert np.array_equal(got, want), k
print("soak ok")
