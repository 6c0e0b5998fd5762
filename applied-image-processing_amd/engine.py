"""Batched AdaIN engine: many frames / views, one style — the shape of the reference's two batch callers
(video/utils.py:327-350 per-frame loop; Style_3DGS/train.py:86-115 per-view loop), which call
``adain_inference`` once per image and re-encode the same style every time.

The engine keeps the packed weights and the style's channel statistics (2 x 512 floats) resident in
HBM and runs encoder -> fused AdaIN/blend -> decoder [-> mask composite] [-> uint8] for a batch of
frames that is already on the GPU.  Sub-batches bound the activation workspace (the full-resolution
activations are 256 B per pixel and frame).  Everything is launched on the current HIP stream.
"""
import torch

from . import runtime as rt


class AdaINEngine:
    def __init__(self, vgg_state_dict, decoder_state_dict, device=None):
        if not torch.cuda.is_available():
            raise rt.AdainHipError("AdaINEngine needs a GPU (no CPU fallback)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.enc = rt.pack_encoder(vgg_state_dict, self.device)
        self.dec = rt.pack_decoder(decoder_state_dict, self.device)
        self.s_mean = self.s_std = None

    def set_style(self, style):
        """style [1,3,hs,ws] (or [1,4,...]: the alpha channel is dropped as in test.py:60-61)."""
        if style.shape[1] == 4:
            style = style[:, :3]
        f = rt.encode(style.to(self.device, torch.float32).contiguous(), self.enc)
        self.s_mean, self.s_std = rt.mean_std(f, True)
        return self

    def features(self, images):
        return rt.encode(images.to(self.device, torch.float32).contiguous(), self.enc)

    def stylize(self, content, alpha=0.5, pmap=None):
        """content [n,3,h,w] on the GPU -> stylised [n,3,8*hc,8*wc].  ``pmap`` [1|n,1,hc,wc] switches to the
        depth-aware blend (test.py:70); otherwise the alpha blend (test.py:80)."""
        assert 0.0 <= alpha <= 1.0
        if self.s_mean is None:
            raise rt.AdainHipError("set_style() first")
        f = rt.encode(content, self.enc)
        c_mean, c_std = rt.mean_std(f, True)
        if pmap is not None:
            g = rt.blend_pmap(f, True, c_mean, c_std, self.s_mean, self.s_std, pmap)
        else:
            g = rt.blend_alpha(f, True, c_mean, c_std, self.s_mean, self.s_std, alpha)
        return rt.decode(g, self.dec)

    def stylize_depth(self, content, depth_maps, offset=0.15, prominence=20):
        """Depth-aware path for a batch: ``depth_maps`` is a list of [h0,w0] GPU tensors, one per frame."""
        assert 0.0 <= offset <= 1.0
        n, _, h, w = content.shape
        hc, wc = rt.encoded_size(h, w)
        p = torch.cat([rt.strength_map(d, hc, wc, offset, prominence) for d in depth_maps])
        return self.stylize(content, pmap=p)

    def composite(self, content, stylized, masks):
        """masks [n|1, 1|3, hm, wm] float -> content*(1-m) + resize(stylized)*m (test.py:222-236)."""
        size = tuple(content.shape[-2:])
        m = rt.resize_nearest(masks.to(self.device, torch.float32), size)
        s = rt.resize_bilinear(stylized, size)
        return rt.mask_composite(content, s, m)

    def to_u8(self, images):
        return rt.quantize_u8(images)
