"""``net.vgg`` / ``net.decoder``: module-level singletons with the reference's state_dict key layout
(Style_3DGS/AdaIN/net.py:6-36 decoder, :38-92 vgg), whose ``forward`` runs the hand-written
gfx950 convolution kernels instead of torch ops.

Both are ``nn.Sequential`` subclasses built from the architecture table (arch.py) so that
``load_state_dict(torch.load("vgg_normalised.pth"))`` / ``decoder.pth``, ``.eval()``, ``.to(device)``
and ``.children()`` behave as on the reference objects; the child modules only hold parameters.

* ``vgg(x)``     : NCHW image [N,3,H,W] -> relu4_1 features shaped [N,512,hc,wc].  The hot path only
                   ever uses the first 31 modules (test.py:185), so that is what ``forward`` computes.
                   The result is ``channels_last`` in memory (the kernels' NHWC activation layout).
* ``decoder(f)`` : features [N,512,hc,wc] (either memory format) -> NCHW image [N,3,8hc,8wc].

The training-only ``Net`` loss wrapper (net.py:95-152) is out of scope.
"""
import torch
import torch.nn as nn

from .. import arch
from .. import runtime as rt
from .function import adaptive_instance_normalization as adain  # noqa: F401  (re-export, net.py:3)
from .function import calc_mean_std  # noqa: F401                                (re-export, net.py:4)


def _modules(table):
    mods = []
    for m in table:
        if m[0] == "conv":
            mods.append(nn.Conv2d(m[1], m[2], (m[3], m[3])))
        elif m[0] == "pad":
            mods.append(nn.ReflectionPad2d((1, 1, 1, 1)))
        elif m[0] == "relu":
            mods.append(nn.ReLU())
        elif m[0] == "pool":
            mods.append(nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True))
        elif m[0] == "up":
            mods.append(nn.Upsample(scale_factor=2, mode="nearest"))
    return mods


class _HipSequential(nn.Sequential):
    _keys = ()

    def __init__(self, table):
        super().__init__(*_modules(table))
        self._packed = None
        self._packed_key = None

    def _pack(self, sd, device):
        raise NotImplementedError

    def packed(self, device):
        """Packed (MFMA-fragment-ordered) weights on ``device``; re-packed when parameters change."""
        params = [p for k in self._keys for p in (self[k].weight, self[k].bias)]
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if self._packed_key != key:
            sd = {f"{k}.{n}": getattr(self[k], n).detach() for k in self._keys for n in ("weight", "bias")}
            self._packed = self._pack(sd, device)
            self._packed_key = key
        return self._packed


class HipVGG(_HipSequential):
    _keys = tuple(rt.ENC_KEYS)

    def __init__(self):
        super().__init__(arch.VGG_MODULES)

    def _pack(self, sd, device):
        return rt.pack_encoder(sd, device)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an image batch [N,3,H,W], got {tuple(x.shape)}")
        x = x.float().contiguous()
        feat = rt.encode(x, self.packed(x.device))       # NHWC [N,hc,wc,512]
        return feat.permute(0, 3, 1, 2)                   # [N,512,hc,wc], channels_last in memory

    def forward_many(self, *xs):
        """``[self(x) for x in xs]`` for image batches of different sizes (content and style of one call, test.py:57,63) in one
        pass over the layers: bit-identical results, one launch per layer instead of one per layer and batch."""
        for x in xs:
            if x.dim() != 4 or x.shape[1] != 3:
                raise ValueError(f"expected an image batch [N,3,H,W], got {tuple(x.shape)}")
        xs = [x.float().contiguous() for x in xs]
        return [f.permute(0, 3, 1, 2) for f in rt.encode_multi(xs, self.packed(xs[0].device))]


class HipDecoder(_HipSequential):
    _keys = tuple(rt.DEC_KEYS)

    def __init__(self):
        super().__init__(arch.DECODER_MODULES)

    def _pack(self, sd, device):
        return rt.pack_decoder(sd, device)

    def forward(self, feat):
        if feat.dim() != 4 or feat.shape[1] != 512:
            raise ValueError(f"expected features [N,512,h,w], got {tuple(feat.shape)}")
        feat = feat.float()
        if feat.is_contiguous(memory_format=torch.channels_last):
            nhwc = feat.permute(0, 2, 3, 1)
        else:
            nhwc = rt.nchw_to_nhwc(feat.contiguous())
        return rt.decode(nhwc, self.packed(feat.device))


decoder = HipDecoder()
vgg = HipVGG()
