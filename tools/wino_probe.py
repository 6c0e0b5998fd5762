#!/usr/bin/env python3
"""Timeline probe of the Winograd conv kernel (conv3x3_wino2_kernel, diagnostic forms 13 / 14): every wave stamps
s_memrealtime (100 MHz) at entry, main-loop start, main-loop end and exit.  Prints per-phase durations, how the two
co-resident workgroups of a CU overlap, and the main-loop time with one workgroup per CU (form 14: LDS-padded), which
shows whether a single wave per SIMD keeps the matrix pipe busy on its own."""
import collections
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


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def probe(cin, cout, h):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w)
    nblk = ((h + 31) // 32) * ((h + 3) // 4) * (cout // 64)
    dbg = torch.zeros(17 * nblk + 8192, dtype=torch.int64, device=dev)
    lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
    flop = 2.0 * h * h * cin * cout * 9
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(50):
            rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 3)
        torch.cuda.synchronize()
    print(f"== {cin}->{cout} @{h}: {nblk} workgroups, ideal main loop per workgroup alone {cin / 8 * 2048 / 2.38e3:.1f} us")
    ref = rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 3)
    for form in (13, 14, 15, 16):
        assert torch.equal(ref, rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form)), form
    for form in (3, 13, 14):
        us = timed(lambda: rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form))
        print(f"form {form}: {us:.1f} us/launch = {flop / us / 1e6:.1f} TF/s (algorithmic)")
    for form in (13, 14):
        dbg.zero_()
        for _ in range(3):
            rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form)
        torch.cuda.synchronize()
        d = dbg.cpu()
        st = d[:16 * nblk].view(nblk, 4, 4).double() / 100.0        # us
        hw = d[16 * nblk:17 * nblk]
        st = st - st[:, :, 0].min()
        ent, ls, le, ex = st[:, :, 0], st[:, :, 1], st[:, :, 2], st[:, :, 3]
        span = ex.max()
        print(f"-- form {form}: kernel span {span:.1f} us")
        for name, v in (("prologue", ls - ent), ("main loop", le - ls), ("epilogue", ex - le)):
            print(f"   {name:9s} us: median {v.median():.2f}  p10 {v.quantile(0.1):.2f}  p90 {v.quantile(0.9):.2f}  max {v.max():.2f}")
        skew = le.max(dim=1).values - le.min(dim=1).values
        print(f"   wave skew at loop end inside a workgroup: median {skew.median():.2f} us, p90 {skew.quantile(0.9):.2f}")
        # matrix-pipe occupancy estimate: fraction of the span during which >= 1 / 2 workgroups of a CU are in their main loop
        xcc = (hw >> 32) & 0xF
        hid = hw & 0xFFFFFFFF
        phys = (((xcc * 8 + ((hid >> 13) & 7)) * 2 + ((hid >> 12) & 1)) * 16 + ((hid >> 8) & 0xF)).tolist()
        bycu = collections.defaultdict(list)
        for i, p in enumerate(phys):
            bycu[p].append(i)
        print(f"   CUs used: {len(bycu)}, workgroups per CU: min {min(len(v) for v in bycu.values())} max {max(len(v) for v in bycu.values())}")
        b0, b1 = ls.min(dim=1).values, le.max(dim=1).values
        tot0 = tot1 = tot2 = 0.0
        for p, blocks in bycu.items():
            ev = []
            for i in blocks:
                ev.append((float(b0[i]), 1))
                ev.append((float(b1[i]), -1))
            ev.sort()
            cur, last = 0, 0.0
            for t, dlt in ev:
                if cur == 0:
                    tot0 += t - last
                elif cur == 1:
                    tot1 += t - last
                else:
                    tot2 += t - last
                last = t
                cur += dlt
            tot0 += float(span) - last
        n = len(bycu) * float(span)
        print(f"   CU time with 0 / 1 / 2+ workgroups in their main loop: {tot0 / n:.3f} / {tot1 / n:.3f} / {tot2 / n:.3f}")
        # lockstep: distribution of main-loop start times modulo nothing -> print entry-time histogram of rounds
        e = ent.min(dim=1).values.sort().values
        print("   entry times (us), every nblk/16-th workgroup:", [round(float(v), 1) for v in e[:: max(1, nblk // 16)]])
    for form in (15, 16):
        dbg.zero_()
        for _ in range(3):
            rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form)
        torch.cuda.synchronize()
        lg = dbg.cpu()[17 * nblk:].view(torch.int32).view(32, 4, 128).long() & 0xFFFFFFFF
        nsteps = min(128, cin // 8 * 4)
        dt = (lg[:, :, 1:nsteps] - lg[:, :, :nsteps - 1]) & 0xFFFFFFFF       # cycles per step, [block][wave][step]
        dt = dt.double()
        print(f"-- form {form} ({'one' if form == 15 else 'two'} workgroups per CU): shader cycles per 8-MFMA step (ideal 512), first 32 workgroups")
        print(f"   all steps: median {dt.median():.0f}  mean {dt.mean():.0f}  p90 {dt.quantile(0.9):.0f}")
        for k in range(8):
            sel = dt[:, :, k::8]
            what = ("A0 wload+patch reads", "A1", "A2 transform adds", "A3 (ends: halo store + barrier)", "B0 halo loads+patch reads", "B1",
                    "B2 transform adds", "B3")[k]
            print(f"   step {k} {what:32s}: median {sel.median():.0f}  mean {sel.mean():.0f}  p90 {sel.quantile(0.9):.0f}")
        w0 = dt[0, 0, :32].tolist()
        print("   workgroup 0 wave 0, first 32 steps:", [int(v) for v in w0])
    lib.adain_debug_set_conv_stamp_buffer(None)


for shape in ((256, 256, 256), (64, 64, 1024)):
    probe(*shape)
