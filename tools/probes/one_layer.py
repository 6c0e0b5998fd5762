#!/usr/bin/env python3
"""Runs one conv layer shape N times with the given Winograd form (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth
cin, cout, h, form, n = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda", 0)
x = torch.from_numpy(synth.uniform_sym(1, (1, h, h, cin), 1.0)).to(dev)
w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
b = torch.zeros(cout, device=dev)
wp = rt.conv3x3_wino_pack(w, form)
for _ in range(n):
    rt.conv3x3_wino(x, wp, b, cout, rt.SRC_DIRECT, True, False, form)
torch.cuda.synchronize()
