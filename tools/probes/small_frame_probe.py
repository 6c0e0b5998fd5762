#!/usr/bin/env python3
"""The reference's real video shape: 256 x 456 frames (adain_inference(content_size=256) on 16:9 video, video/utils.py:264).  Rate of
the job driver over device-resident uint8 frames for several sub-batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
wts = bench.synth_weights()
eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
style = torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev)
cache = {}
for (h, w, n) in ((256, 456, 256), (512, 912, 128), (1080, 1920, 32)):
    dev_frames = torch.stack([synth.frame_u8_torch(7 + k, h, w, dev) for k in range(n)])

    class Store:
        def __len__(self):
            return n

        def __getitem__(self, k):
            return dev_frames[k]

        def block(self, i, j):
            return dev_frames[i:j]

    import applied_image_processing_amd.runtime as rt

    for sub, sched in ((1, rt.SCHEDULE_BATCH), (1, rt.SCHEDULE_LATENCY), (2, rt.SCHEDULE_BATCH), (2, rt.SCHEDULE_LATENCY), (4, rt.SCHEDULE_BATCH),
                       (8, rt.SCHEDULE_BATCH), (16, rt.SCHEDULE_BATCH), (32, rt.SCHEDULE_BATCH)):
        if sub > n:
            continue
        best = 1e9
        with rt.schedule(sched):          # round 6: the latency schedule (cin split of under-filled layers) beside the default
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                jobs.stylize_frames_sharded(eng, Store(), style, sub_batch=sub, style_cache=cache, out_hw=(8 * -(-h // 8), 8 * -(-w // 8)))
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        name = "latency" if sched == rt.SCHEDULE_LATENCY else "batch  "
        print(f"{h}x{w} x {n} frames, sub-batch {sub:2d}, schedule {name}: {n / best:8.1f} frames/s  {n * h * w / 1e6 / best:7.1f} Mpixels/s", flush=True)
