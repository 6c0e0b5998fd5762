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
