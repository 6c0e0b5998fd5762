#!/usr/bin/env python3
"""One wave per SIMD?  Times single conv layers of the persistent F(4,3)xF(2,3) kernel with two workgroups per CU (the product's
grid) and with ONE (ADAIN_W4_WGS=1, diagnostic library): the second number is the matrix-pipe share a single wave per SIMD reaches
with today's instruction mix, every stall and the whole tile epilogue exposed - the regime any work split with more accumulators per
wave (two channel tiles per wave: 192 accumulator registers) would run in.  Run once per setting (the switch is read once)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
PEAK = 157.3e12
print(f"ADAIN_W4_WGS={os.environ.get('ADAIN_W4_WGS', '2')}")
for cin, cout, h in [(64, 64, 1024), (128, 128, 512), (256, 256, 256), (256, 256, 512), (512, 512, 128)]:
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev).clamp_(min=0)       # post-ReLU-like: half zeros
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    for _ in range(5):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    direct = 2.0 * 9 * cin * cout * h * h
    print(f"{cin:4d}->{cout:4d} @{h:4d}^2: {us:8.1f} us  {direct / us / 1e6:7.1f} TF/s algorithmic  matrix pipe {direct / 3 / (us * 1e-6) / PEAK:.3f}", flush=True)
