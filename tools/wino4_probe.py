#!/usr/bin/env python3
"""Per-step timeline of the F(4,3) x F(2,3) Winograd kernel (conv3x3_wino4_kernel diagnostic builds): every wave of the
first 32 workgroups stamps s_memtime at each 8-MFMA double step (ideal 512 cycles).  ADAIN_W4_DIAG=2 pads LDS to one workgroup
per CU (one wave per SIMD), default two."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
lib = rt.lib()
lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]
NAMES = ["A0 patch reads + column combine", "A1 column combine + row transform", "A2 + halo store + barrier",
         "B0 patch reads + 2 halo loads", "B1 transform + 2 halo loads", "B2 + 2 halo loads"]


def probe(cin, cout, h):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    dbg = torch.zeros(32 * 384 // 2 + 64, dtype=torch.int64, device=dev)
    flop = 2.0 * h * h * cin * cout * 9
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50):
            rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"== {cin}->{cout} @{h}: product kernel {us:.1f} us/launch = {flop / us / 1e6:.1f} TF/s (algorithmic)")
    lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
    for _ in range(3):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    lib.adain_debug_set_conv_stamp_buffer(None)
    lg = dbg.cpu()[:32 * 384 // 2].view(torch.int32).view(32, 4, 96).long() & 0xFFFFFFFF
    nsteps = min(96, cin // 8 * 3)
    dt = ((lg[:, :, 1:nsteps] - lg[:, :, :nsteps - 1]) & 0xFFFFFFFF).double()
    print(f"   cycles per 8-MFMA double step (ideal 512): median {dt.median():.0f}  mean {dt.mean():.0f}  p90 {dt.quantile(0.9):.0f}; per stage (6 double steps) mean {dt.mean() * 6:.0f} (ideal 3072)")
    for k in range(6):
        sel = dt[:, :, k::6]
        print(f"   step {k:2d} {NAMES[k]:36s}: median {sel.median():.0f}  mean {sel.mean():.0f}  p90 {sel.quantile(0.9):.0f}")
    print("   workgroup 0 wave 0, first 24 steps:", [int(v) for v in dt[0, 0, :24].tolist()])
    print("   last stage (2 chunks without staging; the last without transform), medians:", [int(dt[:, :, nsteps - 7 + k].median()) for k in range(6)])


for shape in ((256, 256, 256), (64, 64, 1024)):
    probe(*shape)
