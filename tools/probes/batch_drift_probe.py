#!/usr/bin/env python3
"""Do the workgroups of the persistent F(4,3) x F(2,3) kernel drift apart over a longer tile list?  (round 5)

A batch of two 1080p frames in one tile list costs the direct-source mid-network layers 4-6 % per frame and 17 % more L2 misses
(profiles/r04_batch_l2_counters.md); the hypothesis left open in round 4 was DRIFT: the 64 workgroups resident on an XCD walk their
contiguous range with a common stride and are aligned only at launch, so over a list twice as long the window of pixel tiles in flight
(whose halos and the two tile rows' overlap must sit in the XCD's 4 MB L2) widens.  This measures it with the diagnostic library's
per-tile shader-clock stamps (ADAIN_W4_DIAG=4): per XCD, the spread of the 64 workgroups' START TIMES of their k-th tile, in units
of one tile's duration, for the same layer at batch 1 and batch 2."""
import ctypes
import os
import sys

os.environ["ADAIN_W4_DIAG"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _diag  # noqa: F401,E402
import torch

import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

dev = torch.device("cuda", 0)
lib = rt.lib()
lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]
GRID = 512


def probe(cin, cout, h, w, n):
    x = torch.from_numpy(synth.uniform_sym(1, (n, h, w, cin), 1.0)).clamp(min=0).to(dev)
    wt = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = rt.conv3x3_wino_pack(wt, 5)
    items = n * ((w + 31) // 32) * ((h + 7) // 8) * (cout // 32)
    for _ in range(10):
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
    st = dbg.cpu()[:GRID * 1024].view(GRID, 4, 32, 8)[:, 0].double()         # [block][tile][stamp] of wave 0
    per = min(32, items // GRID)                                              # tiles every workgroup has at least
    # a workgroup's clock stamps are comparable with its own only (s_memtime of different CUs differ by large constant offsets): every
    # workgroup of the grid starts with the launch, so progress is measured from each workgroup's own first stamp
    start = st[:, :per, 0] - st[:, :1, 0]                                     # [block][k]: cycles since this workgroup's first tile began
    tile = (st[:, :per, 7] - st[:, :per, 0])
    dur = float(tile[:, 1:per - 1].median()) if per > 2 else float(tile.median())
    print(f"== {cin}->{cout} @ {n} x {h}x{w}: {items} items, {items / GRID:.1f} tiles per workgroup, product kernel {us:.1f} us "
          f"({us / n:.1f} per frame), median tile {dur:.0f} cycles")
    rows, lag = [], []
    for k in range(per):
        sp, lg = [], []
        for xcd in range(8):
            t = start[xcd::8, k]
            sp.append(float(t.max() - t.min()) / dur)
            t0 = float(t.min())                                               # the fastest workgroup of the XCD starts its tile k
            done = (start[xcd::8, :per] <= t0).sum(dim=1)                     # tiles each of the 64 has started by then
            lg.append(float(k + 1 - done.min()))
        rows.append(sum(sp) / 8)
        lag.append(sum(lg) / 8)
    print("   spread of the 64 workgroups' start of tile k on an XCD, in tile durations (mean over the 8 XCDs):")
    print("   " + "  ".join(f"k={k}:{v:.2f}" for k, v in enumerate(rows)))
    print("   tiles the slowest workgroup is behind when the fastest starts tile k (1 tile of lag = 64 list positions = 32 pixel tiles at 2 channel tiles per group):")
    print("   " + "  ".join(f"k={k}:{v:.2f}" for k, v in enumerate(lag)))
    return us / n


for (cin, cout, h, w) in ((256, 256, 270, 480), (128, 128, 540, 960)):
    a = probe(cin, cout, h, w, 1)
    b2 = probe(cin, cout, h, w, 2)
    print(f"   per frame: batch 2 / batch 1 = {b2 / a:.3f}")
