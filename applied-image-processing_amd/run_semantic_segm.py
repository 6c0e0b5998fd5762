"""Command-line front end of the localized (semantic-mask) style transfer with the flag set and defaults of the reference's
Style_3DGS/run_semantic_segm.py (:12-44), so existing invocations keep working:

    python -m applied_image_processing_amd.run_semantic_segm --content c.jpg --style s.jpg [--use_depth]

It calls ``localized.run_localized_style_transfer`` (reference Style_3DGS/localized_style_transfer.py:191-245): background mask ->
``adain_inference(content_mask=..., alpha=1)`` on the MI355X kernels -> Reinhard / PCA / CDF colour transfer of the foreground ->
composite.  Extra flags make it usable offline (the reference downloads DeepLabV3 and MiDaS at run time): ``--mask_npy`` takes a
precomputed background mask ([1,H,W] or [H,W], 1 = background), ``--depth_npy`` a proximity map, ``--vgg`` / ``--decoder`` the
checkpoint paths.  Without ``--mask_npy`` a provider must have been registered (``localized.set_mask_provider``).
"""
import argparse

import numpy as np
import torch

from .localized import run_localized_style_transfer

# (flag, argparse keyword arguments) - names and defaults as in the reference CLI
_REFERENCE_FLAGS = (
    ("--content", dict(type=str, required=True, help="content image file")),
    ("--style", dict(type=str, required=True, help="style image file")),
    ("--output", dict(type=str, default="output", help="directory the results are written to")),
    ("--file_name", dict(type=str, default="stylized", help="name of the intermediate stylised file, extension excluded")),
    ("--use_depth", dict(action="store_true", help="depth-aware stylisation of the background")),
)
_EXTRA_FLAGS = (
    ("--mask_npy", dict(type=str, default=None, help=".npy background mask [1,H,W] or [H,W] (1 = background); replaces the DeepLabV3 estimate")),
    ("--depth_npy", dict(type=str, default=None, help=".npy proximity map [H0,W0]; replaces the MiDaS estimate")),
    ("--vgg", dict(type=str, default="Style_3DGS/AdaIN/models/vgg_normalised.pth", help="encoder state_dict")),
    ("--decoder", dict(type=str, default="Style_3DGS/AdaIN/models/decoder.pth", help="decoder state_dict")),
)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Localized AdaIN style transfer (stylised background, colour-matched foreground) on an MI355X.")
    for flag, kw in _REFERENCE_FLAGS + _EXTRA_FLAGS:
        ap.add_argument(flag, **kw)
    ns = ap.parse_args(argv)
    mask = None
    if ns.mask_npy:
        mask = np.load(ns.mask_npy)
        mask = (mask[None] if mask.ndim == 2 else mask).astype(np.uint8)
    extra = dict(vgg_str=ns.vgg, decoder_str=ns.decoder)
    if ns.depth_npy:
        extra["depth_map"] = torch.from_numpy(np.load(ns.depth_npy).astype(np.float32))
    return run_localized_style_transfer(content_img_path=ns.content, style_img_path=ns.style, output_path=ns.output, file_name=ns.file_name,
                                        use_depth=ns.use_depth, background_mask=mask, **extra)


if __name__ == "__main__":
    main()
