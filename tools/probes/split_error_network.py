#!/usr/bin/env python3
"""What SHORTER accumulation chains are worth to the whole network at trained-like statistics (round 6; the review's item on blocked
accumulation, measured instead of modelled): the latency schedule cuts the cin loop of every under-filled layer into chains of 64
channels (csrc/conv_wino4.hip, cin split) - on a frame small enough that most cin >= 128 layers are split, the two schedules ARE the
A/B of one chain of cin channels against chains of 64.  Prints, per frame size, which layers were split and the relative L2 of the
output image against the fp32 and the float64 oracle under both schedules.   python tools/probes/split_error_network.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import applied_image_processing_amd.arch as arch
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.engine import AdaINEngine
from oracle import adain_oracle as O

ENC = [(64, 64, 0, 1), (64, 128, 0, 0), (128, 128, 0, 1), (128, 256, 0, 0), (256, 256, 0, 0), (256, 256, 0, 0), (256, 256, 0, 1), (256, 512, 0, 0)]
DEC = [(512, 256, 0), (256, 256, 1), (256, 256, 0), (256, 256, 0), (256, 128, 0), (128, 128, 1), (128, 64, 0), (64, 64, 1)]


def split_map(h, w):
    out, ch, cw = [], h, w
    for (ci, co, _, pool) in ENC:
        nb = rt.conv3x3_wino4_split_bytes(1, ch, cw, ci, co)
        out.append(nb // (ch * cw * co * 4) if nb else 1)
        if pool:
            ch, cw = (ch + 1) // 2, (cw + 1) // 2
    for (ci, co, up) in DEC:
        if up:
            ch, cw = 2 * ch, 2 * cw
        nb = rt.conv3x3_wino4_split_bytes(1, ch, cw, ci, co)
        out.append(nb // (ch * cw * co * 4) if nb else 1)
    return out


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


vgg, dec = (synth.to_torch(x) for x in synth.trained_like_state_dicts(0))
v64, d64 = {k: v.double() for k, v in vgg.items()}, {k: v.double() for k, v in dec.items()}
eng = AdaINEngine(vgg, dec, "cuda:0")
s = torch.from_numpy(synth.image(72, 1, 96, 96))
eng.set_style(s.cuda())
print("frame      chains per layer (enc 1_2 .. 4_1 | dec 1 .. 8; 1 = one chain of cin)            batch schedule: vs fp32 / vs f64      latency schedule: vs fp32 / vs f64      oracle fp32 vs f64")
for (h, w) in ((64, 64), (128, 128), (128, 228), (256, 456)):
    c = torch.from_numpy(synth.image(71, 1, h, w))
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg, dec, c, s, 0.5)
        tru = O.style_transfer_simple(v64, d64, c.double(), s.double(), 0.5)
    a = eng.stylize(c.cuda(), 0.5).cpu()
    with rt.schedule(rt.SCHEDULE_LATENCY):
        b = eng.stylize(c.cuda(), 0.5).cpu()
    sm = split_map(h, w)
    print(f"{h:3d}x{w:<3d}    {sm[:8]} | {sm[8:]}    {rel(a, ref):.2e} / {rel(a, tru):.2e}        {rel(b, ref):.2e} / {rel(b, tru):.2e}        {rel(ref, tru):.2e}", flush=True)
