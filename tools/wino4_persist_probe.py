#!/usr/bin/env python3
"""Per-tile phase timeline of the PERSISTENT F(4,3) x F(2,3) kernel (diagnostic library, ADAIN_W4_DIAG=4): wave 0 of every
workgroup stamps the shader clock at tile start, main-loop end, and after the epilogue + next tile's prologue.  Prints the phase
lengths in cycles and, per CU, how the two co-resident workgroups' main loops overlap."""
import collections
import ctypes
import os
import sys

os.environ["ADAIN_W4_DIAG"] = "4"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401,E402
import torch

import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
lib = rt.lib()
lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]
GRID = 512


def probe(cin, cout, h, relu_data=True):
    x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0))
    if relu_data:
        x = x.clamp(min=0)            # post-ReLU-like activations (half zeros), as inside the network
    x = x.to(dev)
    w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(w, 5)
    items = ((h + 31) // 32) * ((h + 7) // 8) * (cout // 32)
    per = items // GRID
    for _ in range(20):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    dbg = torch.zeros(GRID * 1025, dtype=torch.int64, device=dev)
    lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
    for _ in range(2):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 5)
    torch.cuda.synchronize()
    lib.adain_debug_set_conv_stamp_buffer(None)
    d = dbg.cpu()
    st = d[:GRID * 1024].view(GRID, 4, 32, 8)[:, :, :per].double()       # [block][wave][tile][stamp]
    hw = d[GRID * 1024:]
    ideal = cin / 8 * 24 * 64
    print(f"== {cin}->{cout} @{h}^2: {items} items, {per} tiles per workgroup, product kernel {us:.1f} us = {2.0 * h * h * cin * cout * 9 / us / 1e6:.1f} TF/s algorithmic")
    names = ["main loop", "barrier after main loop", "A4^T + P writes + barrier", "P reads + barrier + next halo store",
             "A2 + bias + relu + output stores", "clear accumulators + barrier", "halo load issue + first transform"]
    mid = st[:, :, :-1] if per > 1 else st
    for k, nm in enumerate(names):
        v = (mid[..., k + 1] - mid[..., k]).flatten()
        print(f"   {nm:38s} median {v.median():8.0f}  p10 {v.quantile(0.1):8.0f}  p90 {v.quantile(0.9):8.0f} cycles")
    tile = (mid[..., 7] - mid[..., 0]).flatten()
    print(f"   whole tile median {tile.median():.0f} cycles; MFMA cycles per tile and wave {ideal:.0f} (two waves per SIMD: pipe share {2 * ideal / tile.median():.3f})")
    st0 = st[:, 0]
    xcc = (hw >> 32) & 0xF
    hid = hw & 0xFFFFFFFF
    phys = (((xcc * 8 + ((hid >> 13) & 7)) * 2 + ((hid >> 12) & 1)) * 16 + ((hid >> 8) & 0xF)).tolist()
    bycu = collections.defaultdict(list)
    for i, p in enumerate(phys):
        bycu[p].append(i)
    pairs = [v for v in bycu.values() if len(v) == 2]
    tot = [0.0, 0.0, 0.0]
    for a_, b_ in pairs:
        ev = []
        for i in (a_, b_):
            for t in range(per):
                ev.append((float(st0[i, t, 0]), 1))
                ev.append((float(st0[i, t, 1]), -1))
        ev.sort()
        t_begin = min(float(st0[a_, 0, 0]), float(st0[b_, 0, 0]))
        t_end = max(float(st0[a_, -1, 7]), float(st0[b_, -1, 7]))
        cur, last = 0, t_begin
        for t, dl in ev:
            tot[min(cur, 2)] += t - last
            last = t
            cur += dl
        tot[0] += t_end - last
    # how uneven are the workgroups' walks?  (a workgroup's duration = its last stamp - its first; every workgroup starts with the
    # launch, so the kernel lasts as long as the slowest walk: max / mean - 1 is what a perfect tail balance could win)
    dur = (st0[:, -1, 7] - st0[:, 0, 0])
    print(f"   workgroup walks ({per} tiles): mean {dur.mean():.0f}  p50 {dur.median():.0f}  p90 {dur.quantile(0.9):.0f}  max {dur.max():.0f} cycles; "
          f"max / mean - 1 = {100 * (dur.max() / dur.mean() - 1):.1f} %, p90 / mean - 1 = {100 * (dur.quantile(0.9) / dur.mean() - 1):.1f} %")
    n = sum(tot)
    print(f"   {len(pairs)} CUs with two workgroups: time with 0 / 1 / 2 of them inside a main loop: {tot[0] / n:.3f} / {tot[1] / n:.3f} / {tot[2] / n:.3f}")


for shape in ((64, 64, 1024), (64, 128, 512), (128, 128, 512), (256, 256, 256), (256, 256, 512)):
    probe(*shape)
