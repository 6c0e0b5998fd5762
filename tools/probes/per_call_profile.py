#!/usr/bin/env python3
"""cProfile of the call-by-call path of adain_inference (set_style_cache(False)) in the video caller's shape: where do its
milliseconds go?   python tools/probes/per_call_profile.py"""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image

import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.AdaIN import test as T

root = tempfile.mkdtemp()
torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), os.path.join(root, "vgg.pth"))
torch.save(synth.to_torch(synth.decoder_state_dict(0)), os.path.join(root, "dec.pth"))
ck = dict(vgg_str=os.path.join(root, "vgg.pth"), decoder_str=os.path.join(root, "dec.pth"))
frame = os.path.join(root, "f.jpg")
Image.fromarray((synth.image(7, 1, 270, 480)[0].transpose(1, 2, 0) * 255).astype(np.uint8)).save(frame, quality=95)
style = os.path.join(root, "s.jpg")
Image.fromarray((synth.image(4, 1, 700, 933)[0].transpose(1, 2, 0) * 255).astype(np.uint8)).save(style, quality=95)
depth = torch.from_numpy(synth.smooth_depth(6, 270, 480))
T.set_style_cache(False)


def call():
    with contextlib.redirect_stdout(io.StringIO()):
        T.adain_inference(frame, style, content_size=256, output=os.path.join(root, "o"), use_depth=True, depth_map=depth, depth_offset=0.3, **ck)


for _ in range(3):
    call()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    call()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(35)
print(st.getvalue())
