#!/usr/bin/env python3
"""In-kernel shader-clock probe (MI355X_MICROARCH.md 'DVFS give-back' item 6): runs the diagnostic conv variant
that stamps s_memtime / s_memrealtime around its main loop, after >= 2 s of back-to-back launches on random data.
clock = d(memtime) / d(memrealtime) * 100 MHz, median over workgroups."""
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
cin = cout = 256
h = 256
x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
b = torch.zeros(cout, device=dev)
packed = rt.conv3x3_pack(w)
nblk = (h // 32) * (h // 8) * (cout // 64)
dbg = torch.zeros(7 * nblk, dtype=torch.int64, device=dev)
lib = rt.lib()
lib.adain_debug_set_conv_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.adain_debug_set_conv_stamp_buffer(dbg.data_ptr())
t0 = time.time()
n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        rt.conv3x3(x, packed, b, cout, rt.SRC_DIRECT, True, False, 3)
    torch.cuda.synchronize()
    n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rt.conv3x3(x, packed, b, cout, rt.SRC_DIRECT, True, False, 3)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"warm: {n} launches; steady-state variant 0: {ms*1e3:.1f} us/launch = {2.0*h*h*cin*cout*9/(ms*1e-3)/1e12:.1f} TF/s")
for _ in range(20):
    rt.conv3x3(x, packed, b, cout, rt.SRC_DIRECT, True, False, 10)
torch.cuda.synchronize()
hw = dbg.cpu()[6 * nblk:]
d = dbg.cpu()[:6 * nblk].view(-1, 6).double()
clk = (d[:, 0] / d[:, 1] * 100.0)   # MHz
print(f"in-kernel clock: median {clk.median():.0f} MHz, min {clk.min():.0f}, max {clk.max():.0f}; main loop median {d[:,1].median()/100:.1f} us")
t = d[:, 2:6] / 100.0          # us: entry, loop start, loop end, exit
t0 = t[:, 0].min()
t = t - t0
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f"kernel span (first entry -> last exit): {t[:, 3].max():.1f} us")
print(f"prologue  us: median {pro.median():.2f}  p90 {pro.quantile(0.9):.2f}  max {pro.max():.2f}")
print(f"main loop us: median {loop.median():.2f}  p90 {loop.quantile(0.9):.2f}  max {loop.max():.2f}")
print(f"epilogue  us: median {epi.median():.2f}  p90 {epi.quantile(0.9):.2f}  max {epi.max():.2f}")
order = t[:, 0].argsort()
ent = t[order, 0]
print("block entry times (us), every 64th block in entry order:", [round(float(v), 1) for v in ent[::64]])
ex = t[:, 3].sort().values
print("block exit times (us), every 64th:", [round(float(v), 1) for v in ex[::64]])
lib.adain_debug_set_conv_stamp_buffer(None)

xcc = (hw >> 32) & 0xF
hid = hw & 0xFFFFFFFF
cu = (hid >> 8) & 0xF
sh = (hid >> 12) & 0x1
se = (hid >> 13) & 0x7
phys = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("block -> xcc (first 24):", xcc[:24].tolist())
print("block -> physical cu key (first 16):", phys[:16].tolist())
import collections
first512 = phys[:512].tolist()
cnt = collections.Counter(first512)
print("distinct CUs among first 512 blocks:", len(cnt), "blocks per CU histogram:", collections.Counter(cnt.values()))
pairs = sum(1 for bb in range(256) if phys[bb] == phys[bb + 256])
print("blocks b and b+256 on the same CU:", pairs, "of 256")
# which block index shares the CU with block b among the first 512 (entry order)?
bycu = collections.defaultdict(list)
for bb in range(512):
    bycu[int(phys[bb])].append(bb)
diffs = collections.Counter((v[1] - v[0]) for v in bycu.values() if len(v) == 2)
print("index distance between co-resident first-round blocks:", diffs.most_common(8))
# timing of the staggered experiment variant
for v in (3, 0, 1, 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        rt.conv3x3(x, packed, b, cout, rt.SRC_DIRECT, True, False, v)
    e0.record()
    for _ in range(20):
        rt.conv3x3(x, packed, b, cout, rt.SRC_DIRECT, True, False, v)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"variant {v}: {ms*1e3:.1f} us/launch = {2.0*h*h*cin*cout*9/(ms*1e-3)/1e12:.1f} TF/s")
