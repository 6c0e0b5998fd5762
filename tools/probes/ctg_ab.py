#!/usr/bin/env python3
"""A/B of the persistent walk's channel-tile group (ADAIN_W4_CTG, diagnostic library): time per launch of one layer shape.
    ADAIN_W4_CTG=-1 python tools/probes/ctg_ab.py   # every channel tile in one group (round-1 order)
    ADAIN_W4_CTG=0  python tools/probes/ctg_ab.py   # automatic: weights of a group <= 3 MB"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import _diag  # noqa: F401,E402
import torch
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
for cin, cout, h in ((256, 256, 256), (256, 512, 128), (512, 256, 128), (128, 256, 256), (128, 128, 512), (64, 64, 1024)):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    for _ in range(5):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"CTG={os.environ.get('ADAIN_W4_CTG', '0')} {cin}->{cout} @{h}^2: {us:8.2f} us  {2 * h * h * cin * cout * 9 / us / 1e6:7.1f} TF/s algorithmic")
