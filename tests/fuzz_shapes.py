#!/usr/bin/env python3
"""One-off shape fuzz on the GPU box: random content / style sizes and batches through style_transfer_simple / style_transfer and
the decoder alone, against the CPU oracle: relative L2 <= 1e-4, the parity bar of both weight sets (tests/test_gpu_trained_like.py).
The bar is hard for every case whose content AND style image have at least 48 pixels a side (the reference's defaults are 512 and
256).  Below that - a relu4_1 map of a few dozen positions, where AdaIN divides by the standard deviation of a handful of samples and
amplifies every fp32 path's rounding (the fp32 oracle's own distance from its float64 run reaches 7e-5 there, against 2.3e-5 on
ordinary frames) - a case above the bar is judged by the float64 yardstick AND COUNTED: no further from the float64 oracle than 6 x
the fp32 oracle itself is, and below 5e-4; the summary line lists every such case.  Measured over 800 trained-like cases (both
schedules): 7 above 1e-4 (1.02e-4 ... 3.18e-4, at 2.0 - 5.2 x the oracle's own distance), every one of them with a side below
48 pixels; the worst case with all sides >= 48: 9.0e-5.  With the Kaiming set no case is above 4e-6.  Test infrastructure (it runs the oracle), not collected by pytest:
    python tests/fuzz_shapes.py [n_cases] [seed] [kaiming|trained-like] [batch|latency]
(`latency`: every call under ADAIN_SCHEDULE_LATENCY - the cin split of under-filled launches, which most of these small frames have)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import applied_image_processing_amd.runtime as rt  # noqa: E402
import applied_image_processing_amd.synth as synth  # noqa: E402
from applied_image_processing_amd.AdaIN import net, test as t  # noqa: E402
from oracle import adain_oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KIND = sys.argv[3] if len(sys.argv) > 3 else "kaiming"
SCHEDULE = sys.argv[4] if len(sys.argv) > 4 else "batch"
rt.set_schedule(rt.SCHEDULE_LATENCY if SCHEDULE == "latency" else rt.SCHEDULE_BATCH)
TOL = 1e-4
SMALL_SIDE, YARD_FACTOR, YARD_CAP = 48, 6.0, 5e-4
excused = []
_v, _d = synth.state_dicts(KIND, 0)
vgg_sd, dec_sd = synth.to_torch(_v), synth.to_torch(_d)
net.vgg.load_state_dict(vgg_sd)
net.decoder.load_state_dict(dec_sd)
net.vgg.to("cuda:0")
net.decoder.to("cuda:0")
enc_sd = {k: v for k, v in vgg_sd.items()}
worst = 0.0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([1, 1, 1, 2, 3]))
    h, w = int(rng.integers(9, 330)), int(rng.integers(9, 330))       # below 9 the reference fails too (ReflectionPad2d on a 1-row map)
    hs, ws = int(rng.integers(9, 200)), int(rng.integers(9, 200))
    if rng.random() < 0.15:
        w = int(rng.integers(600, 1100))          # several 32-pixel tiles across, multi-XCD ranges
    c = torch.from_numpy(synth.image(1000 + case, n, h, w))
    s = torch.from_numpy(synth.image(5000 + case, n, hs, ws))
    mode = case % 3
    with torch.no_grad():
        if mode == 0:
            alpha = float(rng.random())
            got = t.style_transfer_simple(net.vgg, net.decoder, c.cuda(), s.cuda(), alpha).cpu()
            ref = O.style_transfer_simple(vgg_sd, dec_sd, c, s, alpha)
        elif mode == 1 and n == 1:
            d = torch.from_numpy(synth.smooth_depth(9000 + case, 2 * h + 3, w + 5))
            got = t.style_transfer(net.vgg, net.decoder, c.cuda(), s.cuda(), d.cuda(), 1.0, 0.2, 15).cpu()
            ref = O.style_transfer(vgg_sd, dec_sd, c, s, d, 1.0, 0.2, 15)
        else:
            hc, wc = max(2, h // 8), max(2, w // 8)
            f = torch.from_numpy(np.maximum(synth.uniform_sym(7000 + case, (n, 512, hc, wc), 3.0), 0).astype(np.float32))
            got = rt.decode(rt.nchw_to_nhwc(f.cuda()), rt.pack_decoder(dec_sd, torch.device("cuda", 0))).cpu()
            ref = O.decode(dec_sd, f)
    rel = float((got - ref).norm() / ref.norm())
    note = ""
    ok = rel <= TOL
    if not ok and mode != 2 and min(h, w, hs, ws) < SMALL_SIDE:
        v64, d64 = {k: v.double() for k, v in vgg_sd.items()}, {k: v.double() for k, v in dec_sd.items()}
        with torch.no_grad():
            tru = (O.style_transfer_simple(v64, d64, c.double(), s.double(), alpha) if mode == 0 else
                   O.style_transfer(v64, d64, c.double(), s.double(), d.double(), 1.0, 0.2, 15))
        mine, floor = float((got.double() - tru).norm() / tru.norm()), float((ref.double() - tru).norm() / tru.norm())
        ok = mine <= YARD_FACTOR * floor and rel <= YARD_CAP
        note = f"   (the fp32 oracle itself is {floor:.2e} from float64, the HIP path {mine:.2e} = {mine / floor:.1f} x)"
        if ok:
            excused.append((case, rel, mine / floor))
            note += f"   A SIDE BELOW {SMALL_SIDE} PIXELS: judged by the float64 yardstick, counted"
    else:
        worst = max(worst, rel)
    flag = note if ok and tuple(got.shape) == tuple(ref.shape) else note + "   <-- FAIL"
    print(f"case {case:3d} mode {mode} n={n} content {h}x{w} style {hs}x{ws}: rel L2 {rel:.2e}{flag}", flush=True)
    flag = "" if ok and tuple(got.shape) == tuple(ref.shape) else "FAIL"
    if flag:
        sys.exit(1)
print(f"{KIND} weights, {SCHEDULE} schedule, {n_cases} cases, worst relative L2 {worst:.2e} (bar {TOL:g}, hard from {SMALL_SIDE} pixels a side); smaller cases above it, judged by the float64 yardstick: "
      f"{len(excused)} {[(c, float(f'{r:.3g}'), float(f'{x:.2g}')) for c, r, x in excused]}, {time.time() - t0:.0f} s")
