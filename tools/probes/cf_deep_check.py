#!/usr/bin/env python3
"""conv_first's two-tiles-ahead variant (diagnostic library, ADAIN_CF_DEEP=1) must give the product kernel's BITS: the same arithmetic,
only the halo loads are requested a tile earlier.  Float and uint8 entry, batches, ragged sizes, one tile per workgroup and many.
    ADAIN_CF_DEEP=1 python tools/probes/cf_deep_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth

assert os.environ.get("ADAIN_CF_DEEP") == "1", "set ADAIN_CF_DEEP=1 (read once, by the diagnostic library)"
dev = torch.device("cuda:0")
vgg = synth.to_torch(synth.vgg_state_dict(0, full=True))
cases = [(1, 1024, 1024), (2, 1080, 1920), (1, 9, 9), (3, 45, 67), (1, 256, 456), (1, 2048, 2048), (1, 8, 3000), (5, 130, 33)]
want = {}
packed = rt.pack_encoder(vgg, dev)
for (n, h, w) in cases:
    x = torch.from_numpy(synth.image(50 + h, n, h, w)).to(dev)
    u8 = (x.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
    want[(n, h, w)] = (rt.encode_relu1_1(x, packed).clone(), rt.encode_relu1_1(u8, packed).clone())
rt.use_library(rt.DIAG_LIB_PATH)
assert rt.is_diag()
packed = rt.pack_encoder(vgg, dev)
for (n, h, w) in cases:
    x = torch.from_numpy(synth.image(50 + h, n, h, w)).to(dev)
    u8 = (x.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
    for rep in range(3):
        a, b = rt.encode_relu1_1(x, packed), rt.encode_relu1_1(u8, packed)
        assert torch.equal(a, want[(n, h, w)][0]) and torch.equal(b, want[(n, h, w)][1]), (n, h, w, rep)
    print(f"{n} x {h} x {w}: float and uint8 entry bit-identical to the product kernel", flush=True)
print("cf_deep_check: ok")
