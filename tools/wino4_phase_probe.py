#!/usr/bin/env python3
"""Phase timeline of the F(4,3) x F(2,3) Winograd kernel (ADAIN_W4_DIAG=3 build): every wave stamps s_memrealtime at
entry, main-loop start, main-loop end and exit.  Prints per-phase durations and how the two co-resident workgroups of
a CU overlap (fraction of CU time with 0 / 1 / 2 workgroups inside their main loops)."""
import collections
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ADAIN_W4_DIAG"] = "3"
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
lib = rt.lib()
lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]


def probe(cin, cout, h):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    nblk = ((h + 31) // 32) * ((h + 7) // 8) * (cout // 32)
    dbg = torch.zeros(17 * nblk, dtype=torch.int64, device=dev)
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
    ideal = cin / 8 * 24 * 64 / 2.38e3
    print(f"== {cin}->{cout} @{h}: {nblk} workgroups, product kernel {us:.1f} us = {flop / us / 1e6:.1f} TF/s; ideal MFMA time per workgroup alone {ideal:.1f} us")
    lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
    for _ in range(3):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    lib.adain_debug_set_conv_stamp_buffer(None)
    d = dbg.cpu()
    st = d[:16 * nblk].view(nblk, 4, 4).double() / 100.0
    hw = d[16 * nblk:17 * nblk]
    st = st - st[:, :, 0].min()
    ent, ls, le, ex = st[:, :, 0], st[:, :, 1], st[:, :, 2], st[:, :, 3]
    span = float(ex.max())
    print(f"   stamped kernel span {span:.1f} us")
    for name, v in (("prologue", ls - ent), ("main loop", le - ls), ("epilogue", ex - le)):
        print(f"   {name:9s} us: median {v.median():.2f}  p10 {v.quantile(0.1):.2f}  p90 {v.quantile(0.9):.2f}")
    xcc = (hw >> 32) & 0xF
    hid = hw & 0xFFFFFFFF
    phys = (((xcc * 8 + ((hid >> 13) & 7)) * 2 + ((hid >> 12) & 1)) * 16 + ((hid >> 8) & 0xF)).tolist()
    bycu = collections.defaultdict(list)
    for i, p in enumerate(phys):
        bycu[p].append(i)
    b0, b1 = ls.min(dim=1).values, le.max(dim=1).values
    tot = [0.0, 0.0, 0.0]
    both = 0.0
    for p, blocks in bycu.items():
        ev = []
        for i in blocks:
            ev.append((float(b0[i]), 1))
            ev.append((float(b1[i]), -1))
        ev.sort()
        cur, last = 0, 0.0
        for t, dl in ev:
            tot[min(cur, 2)] += t - last
            last = t
            cur += dl
        tot[0] += span - last
    n = len(bycu) * span
    f0, f1, f2 = (v / n for v in tot)
    print(f"   CU time with 0 / 1 / 2 workgroups in their main loop: {f0:.3f} / {f1:.3f} / {f2:.3f}")
    loop_sum = float((le - ls).mean()) * nblk / len(bycu)       # main-loop wall time per CU, summed over its workgroups
    mfma = ideal * nblk / len(bycu)
    print(f"   per CU: sum of main-loop times {loop_sum:.1f} us for {mfma:.1f} us of MFMA work per SIMD; if a lone workgroup ran at rate r1 and a pair "
          f"at r2 per workgroup: {mfma:.1f} = r1 * {f1 * span:.1f} + 2 * r2 * {f2 * span:.1f}")


for shape in ((256, 256, 256), (64, 64, 1024), (128, 128, 512)):
    probe(*shape)
