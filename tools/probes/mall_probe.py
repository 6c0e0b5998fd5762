#!/usr/bin/env python3
"""Is it the 256 MB Infinity Cache?  (round 5)  One direct-source 3x3 layer of the product library timed with its
input (a) just written by a producer kernel - what a layer sees inside the network when the schedule runs frame by frame -,
(b) after a 768 MB scratch write has pushed it out of the Infinity Cache - what it sees when the producer layer ran over a whole
batch first -, for activations that fit the cache (132 MB: conv3_x of a 1080p frame) and that do not (265 / 530 MB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
rt.lib()
scratch = torch.empty(768 << 20, dtype=torch.uint8, device=dev)


def layer(cin, cout, h, w, n=1, reps=12):
    x = torch.from_numpy(synth.uniform_sym(1, (n, h, w, cin), 1.0)).clamp(min=0).to(dev)
    src = x.clone()
    wt = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = torch.empty(cin * cout * 24, dtype=torch.float32, device=dev)
    rt._check(rt.lib().adain_conv3x3_wino4_pack(wt.data_ptr(), wp.data_ptr(), cin, cout, rt._stream()), "pack")
    out = torch.empty((n, h, w, cout), dtype=torch.float32, device=dev)

    def conv():
        rt._check(rt.lib().adain_conv3x3_wino(x.data_ptr(), out.data_ptr(), wp.data_ptr(), b.data_ptr(), n, h, w, h, w, cin, cout, 0, 1, 0, 5,
                                              rt._stream()), "conv")

    res = {}
    for mode in ("input just written", "input evicted (768 MB written in between)", "back to back (input last read one launch ago)"):
        ts = []
        for r in range(reps + 3):
            if mode.startswith("input just"):
                x.copy_(src)
            elif mode.startswith("input evicted"):
                x.copy_(src)
                scratch.fill_(r & 255)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            conv()
            e1.record()
            torch.cuda.synchronize()
            if r >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    mb = n * h * w * cin * 4 / 2 ** 20
    print(f"{cin}->{cout} @ {n} x {h}x{w} (input {mb:.0f} MB): " + "; ".join(f"{k}: {v / n:.1f} us per frame" for k, v in res.items()))


for shape in ((256, 256, 270, 480, 1), (256, 256, 270, 480, 2), (128, 128, 540, 960, 1), (128, 128, 540, 960, 2), (64, 64, 1080, 1920, 1),
              (256, 256, 352, 352, 1), (256, 256, 352, 352, 2), (128, 128, 704, 704, 1), (128, 128, 704, 704, 2)):
    layer(*shape)
