"""CLI counterpart of the reference's Style_3DGS/AdaIN/run_depth.py (same flags and defaults, :13-55) on the
MI355X path, plus ``--depth_npy`` / weight-path flags so the depth-aware mode runs offline from files
(the reference fetches MiDaS through torch.hub at run time).

    python -m applied_image_processing_amd.AdaIN.run_depth --content c.jpg --style s.jpg --use_depth --depth_npy d.npy
"""
import argparse

import numpy as np
import torch

from .test import adain_inference


def main(argv=None):
    parser = argparse.ArgumentParser(description="Stylize an image using AdaIN style transfer.")
    parser.add_argument("--content", type=str, required=True, help="Path to the content image.")
    parser.add_argument("--style", type=str, required=True, help="Path to the style image.")
    parser.add_argument("--output", type=str, default="output", help="Output directory.")
    parser.add_argument("--file_name", type=str, default="stylized", help="Output file name without extension.")
    parser.add_argument("--depth_offset", type=float, default=0.15, help="Depth offset for depth-aware style transfer.")
    parser.add_argument("--depth_prominence", type=float, default=20, help="Depth prominence factor.")
    parser.add_argument("--use_depth", action="store_true", help="Enable depth-aware stylization.")
    # additions (not in the reference CLI)
    parser.add_argument("--depth_npy", type=str, default=None, help="Precomputed proximity map [H0,W0] (.npy) instead of MiDaS.")
    parser.add_argument("--vgg", type=str, default="Style_3DGS/AdaIN/models/vgg_normalised.pth")
    parser.add_argument("--decoder", type=str, default="Style_3DGS/AdaIN/models/decoder.pth")
    args = parser.parse_args(argv)

    depth = torch.from_numpy(np.load(args.depth_npy).astype(np.float32)) if args.depth_npy else None
    return adain_inference(
        content_img=args.content,
        style_img=args.style,
        vgg_str=args.vgg,
        decoder_str=args.decoder,
        depth_offset=args.depth_offset,
        depth_prominence=args.depth_prominence,
        output=args.output,
        file_name=args.file_name,
        use_depth=args.use_depth,
        depth_map=depth,
    )


if __name__ == "__main__":
    main()
