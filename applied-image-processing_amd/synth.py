"""Deterministic synthetic weights and inputs (bit-identical on every machine).

The trained checkpoints of the reference are Git-LFS pointers, so parity and
throughput are measured with seeded synthetic weights in the exact reference architecture and
state_dict key layout (Style_3DGS/AdaIN/net.py:6-92).  Only integer hashing and IEEE-exact
float ops (multiply by a power of two, subtract, one correctly-rounded multiply) are used, so
the container that generates the golden fixtures and the GPU box rebuild the same bits; no
transcendental functions, no torch/numpy RNG.

Weights: Kaiming-uniform for ReLU, U(-b, b) with b = sqrt(6 / fan_in); biases U(-0.0866, 0.0866)
(std 0.05).  With these the activations stay O(1) through all 19 layers (relu4_1 mean ~1.4,
std ~2), which keeps PSNR / relative-L2 meaningful.
"""
import math

import numpy as np

from . import arch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    """``n`` float32 values in [0, 1) with 24 random bits each; counter-based on (seed, i)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0xD1342543DE82EF95))[0]
        ctr = np.arange(n, dtype=np.uint64) + base
    bits = _splitmix64(ctr) >> np.uint64(40)
    return bits.astype(np.float32) * np.float32(2.0 ** -24)


def uniform01_bits_torch(seed, n, device="cpu", start=0):
    """The 24 random bits behind ``uniform01(seed, n)[start:start + n]`` as an int64 torch tensor on ``device`` — the same
    counter-based splitmix64, in wrapping int64 arithmetic with logical shifts emulated by masks, so a GPU rebuilds a job's
    synthetic frames bit for bit in microseconds (numpy on the host needs about a second per 1080p frame)."""
    import torch

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = x + _i64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * _i64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)

    with np.errstate(over="ignore"):
        base = int(_splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0xD1342543DE82EF95))[0])
    ctr = torch.arange(start, start + n, dtype=torch.int64, device=device) + _i64(base)
    return lsr(mix(ctr), 40)


def _i64(u):
    """A uint64 constant as the int64 with the same bits."""
    u &= 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


def uniform01_torch(seed, n, device="cpu"):
    """``uniform01`` on a torch device: bit-identical float32 values."""
    import torch

    return uniform01_bits_torch(seed, n, device).to(torch.float32) * (2.0 ** -24)


def frame_u8_torch(seed, h, w, device="cpu", zero_fraction=0.0, zero_seed=0):
    """A synthetic DECODED frame, uint8 [h,w,3] on ``device``: channel c of pixel (y, x) is the top 8 of the 24 random bits of
    element (c, y, x) of ``image(seed, 1, h, w)`` (so ``frame / 255`` is that image quantised the way a decoded picture is).
    ``zero_fraction`` > 0 blanks that share of the pixels (all channels exactly 0: the empty background of a 3DGS training
    view, reference Style_3DGS/train.py:97 ``mask = gt_image_np > 0``), chosen by ``uniform01(zero_seed, h * w) < fraction``."""
    import torch

    bits = uniform01_bits_torch(seed, 3 * h * w, device)
    fr = (bits >> 16).to(torch.uint8).view(3, h, w).permute(1, 2, 0).contiguous()
    if zero_fraction > 0:
        bg = uniform01_torch(zero_seed, h * w, device).view(h, w) < zero_fraction
        fr[bg] = 0
    return fr


def uniform_sym(seed, shape, bound):
    """float32 array U(-bound, bound) of ``shape``."""
    n = int(np.prod(shape))
    u = uniform01(seed, n)
    return ((u - np.float32(0.5)) * np.float32(2.0 * bound)).reshape(shape)


def image(seed, n, h, w, c=3):
    """Synthetic image batch [n, c, h, w] float32 in [0, 1)."""
    return uniform01(seed, n * c * h * w).reshape(n, c, h, w)


def smooth_depth(seed, h, w):
    """Positive smooth field scaled to about [0, 1000] (MiDaS-like inverse depth): 4 low-frequency
    sinusoids + 5 % uniform noise.  float32 [h, w].
    Sinusoids go through float64 numpy sin; last-ulp differences between machines are far below
    the parity tolerance and the map is an *input* (the same array feeds both paths in a test)."""
    yy = np.arange(h, dtype=np.float64)[:, None] / max(h, 1)
    xx = np.arange(w, dtype=np.float64)[None, :] / max(w, 1)
    f = (
        np.sin(2 * math.pi * (1.0 * yy + 0.5 * xx))
        + 0.7 * np.sin(2 * math.pi * (0.5 * yy - 1.5 * xx) + 1.0)
        + 0.5 * np.sin(2 * math.pi * (2.0 * yy + 1.0 * xx) + 2.0)
        + 0.3 * np.sin(2 * math.pi * (3.0 * xx) + 0.5)
    )
    f = (f - f.min()) / (f.max() - f.min())
    noise = uniform01(seed, h * w).reshape(h, w).astype(np.float64)
    return ((0.95 * f + 0.05 * noise) * 1000.0).astype(np.float32)


def _conv_params(seed, net_id, idx, cin, cout, k):
    fan_in = cin * k * k
    wb = math.sqrt(6.0 / fan_in)
    s = (seed * 1000003 + net_id * 1009 + idx) * 2
    w = uniform_sym(s, (cout, cin, k, k), wb)
    b = uniform_sym(s + 1, (cout,), 0.05 * math.sqrt(3.0))
    return w, b


def _state_dict(mods, net_id, seed):
    sd = {}
    for i, m in enumerate(mods):
        if m[0] == "conv":
            w, b = _conv_params(seed, net_id, i, m[1], m[2], m[3])
            sd[f"{i}.weight"] = w
            sd[f"{i}.bias"] = b
    return sd


def vgg_state_dict(seed=0, full=True):
    """numpy state_dict for the encoder.  ``full`` = all 17 convs (needed by a strict
    ``load_state_dict`` into the 53-module reference ``net.vgg``); otherwise only the 10 convs up
    to relu4_1."""
    mods = arch.VGG_MODULES if full else arch.VGG_MODULES[: arch.ENCODER_CUT]
    return _state_dict(mods, 1, seed)


def decoder_state_dict(seed=0):
    return _state_dict(arch.DECODER_MODULES, 2, seed)


def to_torch(sd):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# ---- a second weight set: "trained-like" statistics -------------------------------------------------------------------------------
# The checkpoint the reference really loads (test.py:183-185; an LFS pointer here) is the "normalised" VGG-19 of the AdaIN release:
# conv0 carries Caffe-style preprocessing (x255, RGB -> BGR, mean subtraction: pixel values near +-100 enter conv1_1), filters have
# non-zero means and a share of zero-sum (edge-detector) filters, and every channel was rescaled so that its post-ReLU activation
# averages 1 over images - activations with a large DC part instead of the zero-mean O(1) noise the Kaiming set produces.  That
# regime is what stresses the folded first layer (conv0 into conv1_1: the x255 terms cancel against a folded bias of order 100 x
# sum|w|) and the F(4,3) x F(2,3) transforms (a DC of 1 under zero-sum filters).  This set rebuilds it from the integer PRNG:
#   conv0            W = 255 * [[0,0,1],[0,1,0],[1,0,0]], b = (-103.939, -116.779, -123.68)
#   every 3x3 conv   Kaiming-uniform base; output channel o is, by o % 4: 0 a "blob" filter (every weight + 6 / fan_in: a DC gain of 6
#                    on mean-1 inputs), 1 a spatial edge detector (each 3x3 slice minus its own mean: zero sum per input channel),
#                    2 an opponent filter (the whole filter minus its mean), 3 the base minus 1 / fan_in (sparse activations);
#   normalisation    layer by layer, channel o of (w, b) is multiplied by 1 / (its post-ReLU mean over two calibration images), as the
#                    normalised VGG was made; the decoder the same way on AdaIN(content, style) features, its last layer scaled to
#                    an image of mean 0.5 / std 0.22 per colour.
# The calibration forward runs in float64 and every scale is rounded to 9 significant bits before it multiplies the float32 weights
# (one correctly rounded multiply): last-bit differences between machines' float64 convolutions cannot move a scale, so the
# container that writes tests/golden/case_g.npz and the GPU box rebuild the same weights bit for bit.
_TL_CACHE = {}


def _round_bits(x, bits=9):
    """float64 array rounded to ``bits`` significant bits."""
    m, e = np.frexp(np.asarray(x, dtype=np.float64))
    return np.ldexp(np.round(m * (1 << bits)) / (1 << bits), e)


def _calibration_images(seed):
    """Two 64 x 64 RGB images in [0, 1): three quarters uniform noise - what the parity tests and the bench feed, so the channel means
    sit near 1 THERE - and one quarter smooth field."""
    out = []
    for k in range(2):
        sm = smooth_depth(seed * 7919 + 31 + k, 64, 64).astype(np.float64) / 1000.0
        rgb = np.stack([sm, sm[::-1, :], sm[:, ::-1]])
        out.append(0.25 * rgb + 0.75 * image(seed * 7919 + 41 + k, 1, 64, 64)[0].astype(np.float64))
    return np.stack(out)


def _shaped_filter(seed, net_id, idx, cin, cout):
    w, b = _conv_params(seed, net_id + 10, idx, cin, cout, 3)
    fan_in = cin * 9
    w = w.copy()
    w[0::4] += np.float32(6.0 / fan_in)
    w[1::4] -= w[1::4].mean(axis=(2, 3), keepdims=True, dtype=np.float32)
    w[2::4] -= w[2::4].mean(axis=(1, 2, 3), keepdims=True, dtype=np.float32)
    w[3::4] -= np.float32(1.0 / fan_in)
    return w, (b * np.float32(2.0)).astype(np.float32)


def trained_like_state_dicts(seed=0):
    """``(vgg_state_dict (all 17 convs), decoder_state_dict)`` as numpy, reference key layout: see the block comment above."""
    if seed in _TL_CACHE:
        return _TL_CACHE[seed]
    import torch
    import torch.nn.functional as F

    def conv(x, w, b):
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), torch.from_numpy(w.astype(np.float64)), torch.from_numpy(b.astype(np.float64)))

    def normalised(x, w, b):
        """(w, b) with every output channel scaled to a post-ReLU mean of 1 on x."""
        mean = F.relu(conv(x, w, b)).mean(dim=(0, 2, 3)).numpy()
        s = _round_bits(1.0 / np.maximum(mean, 0.25 * np.median(mean))).astype(np.float32)     # (rarely firing channels: gain capped)
        return w * s[:, None, None, None], b * s

    vgg = vgg_state_dict(seed, full=True)                   # the convs behind relu4_1 stay Kaiming: net.vgg[:31] never runs them
    vgg["0.weight"] = (np.float32(255.0) * np.eye(3, dtype=np.float32)[::-1]).reshape(3, 3, 1, 1).copy()
    vgg["0.bias"] = np.array([-103.939, -116.779, -123.68], dtype=np.float32)
    x = torch.from_numpy(_calibration_images(seed))
    x = F.conv2d(x, torch.from_numpy(vgg["0.weight"].astype(np.float64)), torch.from_numpy(vgg["0.bias"].astype(np.float64)))
    for i, m in enumerate(arch.VGG_MODULES[: arch.ENCODER_CUT]):
        if m[0] == "pool":
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
        elif m[0] == "conv" and m[3] == 3:
            w, b = normalised(x, *_shaped_filter(seed, 1, i, m[1], m[2]))
            vgg[f"{i}.weight"], vgg[f"{i}.bias"] = w, b
            x = F.relu(conv(x, w, b))
    # decoder: calibrated on AdaIN(content_f, style_f) of the two images (function.py:15-23)
    mu, sd = x.mean(dim=(2, 3), keepdim=True), (x.var(dim=(2, 3), keepdim=True) + 1e-5).sqrt()
    x = (x - mu) / sd * sd.flip(0) + mu.flip(0)
    dec = {}
    convs = arch.conv_indices(arch.DECODER_MODULES)
    for i, m in enumerate(arch.DECODER_MODULES):
        if m[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif m[0] == "conv":
            w, b = _shaped_filter(seed, 2, i, m[1], m[2])
            if i != convs[-1]:
                w, b = normalised(x, w, b)
                x = F.relu(conv(x, w, b))
            else:                                           # 64 -> 3, no ReLU: an image of mean 0.5, std 0.22 per colour
                y = conv(x, w, np.zeros_like(b))
                s = _round_bits(0.22 / y.std(dim=(0, 2, 3)).numpy()).astype(np.float32)
                w = w * s[:, None, None, None]
                b = _round_bits(0.5 - s.astype(np.float64) * y.mean(dim=(0, 2, 3)).numpy()).astype(np.float32)
            dec[f"{i}.weight"], dec[f"{i}.bias"] = w, b
    _TL_CACHE[seed] = (vgg, dec)
    return vgg, dec


def state_dicts(kind="kaiming", seed=0):
    """``(vgg_state_dict (full), decoder_state_dict)`` of a named weight set: "kaiming" (zero-mean, O(1) activations: the set every
    round-1..4 fixture uses) or "trained-like" (see above)."""
    if kind == "kaiming":
        return vgg_state_dict(seed, full=True), decoder_state_dict(seed)
    if kind == "trained-like":
        return trained_like_state_dicts(seed)
    raise ValueError(f"unknown weight set {kind!r}")
