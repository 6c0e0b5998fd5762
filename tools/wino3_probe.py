#!/usr/bin/env python3
"""Timeline probe of the persistent Winograd kernel (conv3x3_wino3_kernel, diagnostic form 17): wave 0 of every
workgroup stamps s_memrealtime at main-loop start / end and epilogue end of its first 16 tiles.  Shows per-phase
durations and how the two workgroups of a CU overlap (env ADAIN_W3_STAGGER selects the start offset)."""
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
    items = ((h + 31) // 32) * ((h + 3) // 4) * (cout // 64)
    grid = min(512, (items + 7) // 8 * 8)
    dbg = torch.zeros(grid * 50, dtype=torch.int64, device=dev)
    lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
    flop = 2.0 * h * h * cin * cout * 9
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50):
            rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 4)
        torch.cuda.synchronize()
    print(f"== {cin}->{cout} @{h}: {items} tiles, grid {grid}, ideal MFMA time per tile {cin / 8 * 2048 / 2.38e3:.1f} us")
    for form in (3, 4, 17):
        us = timed(lambda: rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form))
        print(f"form {form}: {us:.1f} us/launch = {flop / us / 1e6:.1f} TF/s (algorithmic)")
    dbg.zero_()
    for _ in range(3):
        rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, 17)
    torch.cuda.synchronize()
    d = dbg.cpu()
    st = d[:grid * 48].view(grid, 16, 3).double() / 100.0
    hw = d[grid * 48:grid * 49]
    valid = st[:, :, 0] > 0
    t0v = st[:, :, 0][valid].min()
    st = st - t0v
    nt = int(valid[0].sum())
    loop = (st[:, :, 1] - st[:, :, 0])[valid]
    epi = (st[:, :, 2] - st[:, :, 1])[valid]
    gap = (st[:, 1:nt, 0] - st[:, :nt - 1, 2]).flatten() if nt > 1 else torch.zeros(1, dtype=torch.float64)
    per = (st[:, 1:nt, 0] - st[:, :nt - 1, 0]).flatten() if nt > 1 else torch.zeros(1, dtype=torch.float64)
    print(f"   tiles stamped per workgroup: {nt}; span {st[:, :, 2][valid].max():.1f} us")
    for name, v in (("main loop", loop), ("epilogue", epi), ("re-entry", gap), ("tile period", per)):
        print(f"   {name:11s} us: median {v.median():.2f}  p10 {v.quantile(0.1):.2f}  p90 {v.quantile(0.9):.2f}")
    xcc = (hw >> 32) & 0xF
    hid = hw & 0xFFFFFFFF
    phys = (((xcc * 8 + ((hid >> 13) & 7)) * 2 + ((hid >> 12) & 1)) * 16 + ((hid >> 8) & 0xF)).tolist()
    bycu = collections.defaultdict(list)
    for i, p in enumerate(phys):
        bycu[p].append(i)
    pairs = sum(1 for bb in range(grid // 2) if phys[bb] == phys[bb + grid // 2])
    print(f"   CUs used {len(bycu)}; workgroups b and b+{grid // 2} on the same CU: {pairs} of {grid // 2}")
    span = float(st[:, :, 2][valid].max())
    tot = [0.0, 0.0, 0.0]
    for p, blocks in bycu.items():
        ev = []
        for i in blocks:
            for k in range(nt):
                ev.append((float(st[i, k, 0]), 1))
                ev.append((float(st[i, k, 1]), -1))
        ev.sort()
        cur, last = 0, 0.0
        for t, dl in ev:
            tot[min(cur, 2)] += t - last
            last = t
            cur += dl
        tot[0] += span - last
    n = len(bycu) * span
    print(f"   CU time with 0 / 1 / 2 workgroups in their main loop: {tot[0] / n:.3f} / {tot[1] / n:.3f} / {tot[2] / n:.3f}")
    cu0 = next(iter(bycu.values()))
    for i in cu0[:2]:
        print(f"   workgroup {i} (first CU): loop start/end, epilogue end (us):", [tuple(round(float(v), 1) for v in st[i, k]) for k in range(min(nt, 4))])
    ck = d[grid * 49:grid * 50]
    ck = ck[ck != 0]
    if len(ck):
        mhz = ((ck >> 32) & 0xFFFFFFFF).double() / (ck & 0xFFFFFFFF).double() * 100.0
        print(f"   in-kernel shader clock over the second tile's main loop: median {mhz.median():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
    lib.adain_debug_set_conv_stamp_buffer(None)


for shape in ((256, 256, 256), (64, 64, 1024)):
    probe(*shape)
