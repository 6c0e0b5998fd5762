"""Command-line front end of the MI355X AdaIN path with the flag set and defaults of the reference's
Style_3DGS/AdaIN/run_depth.py (:13-55), so existing invocations keep working:

    python -m applied_image_processing_amd.AdaIN.run_depth --content c.jpg --style s.jpg [--use_depth]

Extra flags make the depth-aware mode usable offline (the reference pulls MiDaS through torch.hub at run time):
``--depth_npy`` takes a precomputed proximity map, ``--vgg`` / ``--decoder`` the checkpoint paths.
"""
import argparse

import numpy as np
import torch

from .test import adain_inference

# (flag, argparse keyword arguments) — names and defaults as in the reference CLI
_REFERENCE_FLAGS = (
    ("--content", dict(type=str, required=True, help="content image file")),
    ("--style", dict(type=str, required=True, help="style image file")),
    ("--output", dict(type=str, default="output", help="directory the result is written to")),
    ("--file_name", dict(type=str, default="stylized", help="result file name, extension excluded")),
    ("--depth_offset", dict(type=float, default=0.15, help="cap of the strength map is 1 - offset (depth-aware mode)")),
    ("--depth_prominence", dict(type=float, default=20, help="slope of the sigmoid applied to the proximity map")),
    ("--use_depth", dict(action="store_true", help="blend by the depth-proximity map instead of a global alpha")),
)
_EXTRA_FLAGS = (
    ("--depth_npy", dict(type=str, default=None, help=".npy proximity map [H0,W0]; replaces the MiDaS estimate")),
    ("--vgg", dict(type=str, default="Style_3DGS/AdaIN/models/vgg_normalised.pth", help="encoder state_dict")),
    ("--decoder", dict(type=str, default="Style_3DGS/AdaIN/models/decoder.pth", help="decoder state_dict")),
)


def main(argv=None):
    ap = argparse.ArgumentParser(description="AdaIN style transfer of one image on an MI355X.")
    for flag, kw in _REFERENCE_FLAGS + _EXTRA_FLAGS:
        ap.add_argument(flag, **kw)
    ns = ap.parse_args(argv)
    proximity = None
    if ns.depth_npy:
        proximity = torch.from_numpy(np.load(ns.depth_npy).astype(np.float32))
    return adain_inference(ns.content, ns.style, vgg_str=ns.vgg, decoder_str=ns.decoder, depth_offset=ns.depth_offset,
                           depth_prominence=ns.depth_prominence, output=ns.output, file_name=ns.file_name,
                           use_depth=ns.use_depth, depth_map=proximity)


if __name__ == "__main__":
    main()
