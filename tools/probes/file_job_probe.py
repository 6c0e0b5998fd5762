#!/usr/bin/env python3
"""The reference's loop shape end to end - JPEG file in, stylised JPEG file out (video/utils.py:341-366, train.py:86-115) - through
the job driver: N 1080p JPEG frames on disk -> PIL decode on the feeder's worker pool -> pinned uint8 upload -> kernels -> async
D2H -> JPEG encode + write on the sink's pool.  Prints frames/s for a few pool sizes next to the HBM-resident rate."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n, h, w = int(os.environ.get("N", 96)), 1080, 1920
wts = bench.synth_weights()
eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
style = torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev)
cache = {}
tmp = tempfile.mkdtemp(prefix="adain_files_")
src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
os.makedirs(src); os.makedirs(dst)
smooth = (np.indices((h, w)).sum(0) % 256).astype(np.uint8)
for k in range(n):          # photo-like content (smooth gradients + noise) so that the JPEGs have realistic sizes / decode cost
    fr = synth.frame_u8_torch(7 + k, h, w, dev).cpu().numpy()
    img = (0.75 * np.stack([np.roll(smooth, 37 * k + 11 * c, 1) for c in range(3)], -1) + 0.25 * fr).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(src, f"frame_{k:04d}.jpg"), quality=90)
names = sorted(os.listdir(src))
print(f"{n} JPEG frames, {sum(os.path.getsize(os.path.join(src, f)) for f in names) / n / 1e6:.2f} MB each", flush=True)


class Files:
    def __len__(self):
        return n

    def __getitem__(self, k):
        return Image.open(os.path.join(src, names[k])).convert("RGB")


t0 = time.perf_counter(); [np.asarray(Files()[k]) for k in range(8)]; dec = (time.perf_counter() - t0) / 8
print(f"PIL decode of one frame on one core: {dec * 1e3:.1f} ms", flush=True)
res = bench.FrameStore(4, n, 0, n, h, w, dev, host=False)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    jobs.stylize_frames_sharded(eng, res, style, sub_batch=2, style_cache=cache, out_hw=(h, w))
    torch.cuda.synchronize(); t_res = time.perf_counter() - t0
print(f"HBM-resident, no files: {n / t_res:7.1f} frames/s", flush=True)
for workers in (4, 8, 12):
    for rep in range(2):
        sink = jobs.FileSink(dev, workers=workers)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, info = jobs.stylize_frames_sharded(eng, Files(), style, sub_batch=2, style_cache=cache, out_hw=(h, w), gather=False, fetch_workers=workers,
                                              sink=lambda i, j, u8: sink.write(u8, [os.path.join(dst, names[k]) for k in range(i, j)]))
        t_gpu = time.perf_counter() - t0
        sink.close()
        t_all = time.perf_counter() - t0
    print(f"JPEG in -> JPEG out, {workers:2d} decode + {workers:2d} encode threads: {n / t_all:7.1f} frames/s (kernels done after {t_gpu:.2f} s, files after {t_all:.2f} s; "
          f"feeder {info['feeder']})", flush=True)
assert len(os.listdir(dst)) == n
