#!/usr/bin/env python3
"""Round 5: the fp32 error of single F(4,3) x F(2,3) layers ON THE GPU, at trained-like statistics, against a float64 convolution -
next to torch's fp32 direct convolution on the host (the reference's arithmetic) and to the CPU model of tools/probes/wino_error_model.py.
Inputs are the float64 activations of the trained-like encoder on a 64 x 64 noise image (computed here with torch's double conv2d),
rounded to fp32; every layer is fed the SAME fp32 input on both paths.  Round 6: a third column, the same layer split along cin (the
latency schedule's form, adain_conv3x3_wino4_split: S chains of cin / S channels, partial sums added in fp32 in a fixed order) where
the launch is small enough to be split.   python tools/probes/wino_error_gpu.py [H W]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import applied_image_processing_amd.arch as arch  # noqa: E402
import applied_image_processing_amd.runtime as rt  # noqa: E402
import applied_image_processing_amd.synth as synth  # noqa: E402

rt.lib()
vgg, _ = synth.trained_like_state_dicts(0)


def conv(x, w, b):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
x = torch.from_numpy(synth.image(11, 1, H, W)).double()
x = F.conv2d(x, torch.from_numpy(vgg["0.weight"]).double(), torch.from_numpy(vgg["0.bias"]).double())
print("layer (state_dict index)   cin->cout @ map      direct fp32 (torch CPU)   F(4,3)xF(2,3) fp32 (GPU)   ratio     split along cin: S, error, ratio to direct, unsplit / split")
for i, m in enumerate(arch.VGG_MODULES[: arch.ENCODER_CUT]):
    if m[0] == "pool":
        x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
    elif m[0] == "conv" and m[3] == 3:
        w, b = torch.from_numpy(vgg[f"{i}.weight"]), torch.from_numpy(vgg[f"{i}.bias"])
        truth = F.relu(conv(x, w.double(), b.double()))
        if m[1] >= 32:                     # the generic layers (conv1_1 is the folded first layer)
            x32 = x.float()
            direct = F.relu(conv(x32, w, b))
            packed = torch.empty(m[1] * m[2] * 24, dtype=torch.float32, device="cuda")
            rt._check(rt.lib().adain_conv3x3_wino4_pack(w.cuda().contiguous().data_ptr(), packed.data_ptr(), m[1], m[2], rt._stream()), "pack")
            got = rt.conv3x3_wino(x32.cuda().permute(0, 2, 3, 1).contiguous(), packed, b.cuda(), m[2], rt.SRC_DIRECT, True, False, 5)
            got = got.permute(0, 3, 1, 2).cpu()
            d, g = rel(direct, truth), rel(got, truth)
            nb = rt.conv3x3_wino4_split_bytes(1, x.shape[2], x.shape[3], m[1], m[2])
            tail = "     (not split at this size)"
            if nb:
                S = nb // (x.shape[2] * x.shape[3] * m[2] * 4)
                sp = rt.conv3x3_wino4_split(x32.cuda().permute(0, 2, 3, 1).contiguous(), packed, b.cuda(), m[2], rt.SRC_DIRECT, True, False)
                e = rel(sp.permute(0, 3, 1, 2).cpu(), truth)
                tail = f"     S = {S} ({m[1] // S} channels per chain)  {e:.2e}  {e / d:.1f}  {g / e:.2f}"
            print(f"   {i:2d}                      {m[1]:3d}->{m[2]:3d} @ {x.shape[2]:3d}x{x.shape[3]:<3d}      {d:.2e}                 {g:.2e}                 {g / d:.1f}{tail}")
        x = truth
