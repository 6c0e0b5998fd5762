#!/usr/bin/env python3
"""A/B timing of a diagnostic (possibly timing-only) build of the F(4,3) x F(2,3) kernel against the one-tile product kernel:
ADAIN_W4_DIAG=<n> ADAIN_W4_PERSIST=0 python tools/probes/diag_ab.py   (the stamp buffer being set selects the diagnostic build)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth
dev = torch.device("cuda", 0)
lib = rt.lib(); lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]
for cin, cout, h in ((256, 256, 256), (64, 64, 1024), (128, 128, 512)):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    flop = 2.0 * h * h * cin * cout * 9
    res = []
    for use_dbg in (False, True, False, True):
        lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr() if use_dbg else None)
        for _ in range(100): rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
        e1.record(); torch.cuda.synchronize()
        res.append(flop / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e12)
    lib.adain_debug_set_conv_stamp_buffer(None)
    print(cin, cout, h, "product / diag alternating TF/s:", [round(v, 1) for v in res])
