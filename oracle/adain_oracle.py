"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement (torch fp32, functional ops) of the
reference's AdaIN inference path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``applied-image-processing_amd/``) never does and fails loudly when the HIP library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the reference's own unmodified ``function.py`` / ``net.py`` /
``test.py`` in the build container (``tests/golden/make_golden.py``; the reference has no tests
or fixtures of its own — SURVEY.md section 4).  Parts of the reference that need torchvision
(``test_transform``, ``save_image``) cannot be imported there; their restatements
(``resize_size``, ``quantize_u8``) are pinned by construction-level known answers only and are
marked "parity unpinned" below.

All file:line citations are relative to /root/reference/Style_3DGS/AdaIN/.
"""
import torch
import torch.nn.functional as F

# state_dict indices of the convs, in module order (net.py:38-69 encoder cut at 31, net.py:6-36 decoder)
ENC_CONVS = [(0, 1, False), (2, 3, True), (5, 3, True), ("pool",), (9, 3, True), (12, 3, True), ("pool",),
             (16, 3, True), (19, 3, True), (22, 3, True), (25, 3, True), ("pool",), (29, 3, True)]
DEC_CONVS = [(1, True), ("up",), (5, True), (8, True), (11, True), (14, True), ("up",), (18, True), (21, True),
             ("up",), (25, True), (28, False)]


def _conv3x3_reflect(x, w, b):
    # ReflectionPad2d((1,1,1,1)) + Conv2d(k=3)  (net.py:7-8 and every later pair)
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def encode(sd, x):
    """net.vgg[:31] (net.py:38-69; cut at test.py:185): conv0 1x1, then 9 x [pad, conv3x3, relu]
    with MaxPool2d(2, 2, ceil_mode=True) after relu1_2, relu2_2, relu3_4.  x: [N,3,H,W]."""
    for e in ENC_CONVS:
        if e[0] == "pool":
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
            continue
        idx, k, relu = e
        w, b = sd[f"{idx}.weight"], sd[f"{idx}.bias"]
        x = F.conv2d(x, w, b) if k == 1 else _conv3x3_reflect(x, w, b)
        if relu:
            x = F.relu(x)
    return x


def decode(sd, x):
    """net.decoder (net.py:6-36): 9 x [pad, conv3x3] with ReLU after all but the last and a
    nearest 2x Upsample after convs 1, 5 and 7 (module indices 3, 16, 23)."""
    for e in DEC_CONVS:
        if e[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            continue
        idx, relu = e
        x = _conv3x3_reflect(x, sd[f"{idx}.weight"], sd[f"{idx}.bias"])
        if relu:
            x = F.relu(x)
    return x


def calc_mean_std(feat, eps=1e-5):
    """function.py:4-12 — per (n, c) mean and sqrt(unbiased var + eps) over H*W."""
    assert feat.dim() == 4
    n, c = feat.shape[:2]
    flat = feat.reshape(n, c, -1)
    var = flat.var(dim=2) + eps
    return flat.mean(dim=2).view(n, c, 1, 1), var.sqrt().view(n, c, 1, 1)


def adaptive_instance_normalization(content_feat, style_feat):
    """function.py:15-23."""
    assert content_feat.shape[:2] == style_feat.shape[:2]
    s_mean, s_std = calc_mean_std(style_feat)
    c_mean, c_std = calc_mean_std(content_feat)
    normalized = (content_feat - c_mean) / c_std
    return normalized * s_std + s_mean


def compute_stylization_strength_map(depth_map, encoder_size, offset=0.15, prominence=20):
    """test.py:119-150 — bicubic resize, min-max normalise, centre on the mean, sigmoid with
    slope ``prominence``, clamp to 1 - offset.  depth_map [H0,W0] -> [1,1,Hc,Wc]."""
    hc, wc = encoder_size
    p = F.interpolate(depth_map[None, None], size=(hc, wc), mode="bicubic", align_corners=False)
    lo, hi = p.min(), p.max()
    if not bool(hi > lo):
        return torch.zeros_like(p)
    p = (p - lo) / (hi - lo)
    p = p - p.mean()
    p = 1.0 / (1.0 + torch.exp(-prominence * p))
    return torch.clamp(p, max=1.0 - offset)


def style_transfer_simple(vgg_sd, dec_sd, content, style, alpha=0.5):
    """test.py:74-81."""
    assert 0.0 <= alpha <= 1.0
    content_f = encode(vgg_sd, content)
    style_f = encode(vgg_sd, style)
    feat = adaptive_instance_normalization(content_f, style_f)
    feat = feat * alpha + content_f * (1 - alpha)
    return decode(dec_sd, feat)


def style_transfer(vgg_sd, dec_sd, content, style, depth_map, alpha=1.0, offset=0.15, prominence=20):
    """test.py:52-71 — depth-aware blend; ``alpha`` is only asserted, never used (test.py:55,70)."""
    assert 0.0 <= alpha <= 1.0
    assert 0.0 <= offset <= 1.0
    content_f = encode(vgg_sd, content)
    if style.shape[1] == 4:
        style = style[:, :3]
    style_f = encode(vgg_sd, style)
    hc, wc = content_f.shape[2:]
    p = compute_stylization_strength_map(depth_map, (hc, wc), offset, prominence)
    t = adaptive_instance_normalization(content_f, style_f)
    feat = t * (1 - p) + content_f * p
    return decode(dec_sd, feat)


def mask_composite(content, output_img, mask):
    """test.py:222-236 — mask [C',H',W'] (any numeric/bool) -> float, unsqueeze(0), nearest resize to
    the content size; stylised output bilinear-resized (align_corners=False) to the content size;
    content * (1 - m) + out * m."""
    m = mask.float().unsqueeze(0)
    m = F.interpolate(m, size=content.shape[-2:], mode="nearest")
    out = F.interpolate(output_img, size=content.shape[-2:], mode="bilinear", align_corners=False)
    return content * (1.0 - m) + out * m


def quantize_u8(img):
    """torchvision.utils.save_image quantiser (test.py:243-244; torchvision 0.13 utils.py):
    ``img.mul(255).add_(0.5).clamp_(0, 255).to(uint8)`` on a [N,3,H,W] tensor -> [N,H,W,3] u8.
    PARITY UNPINNED by import (torchvision absent); pinned by known answers in the tests."""
    return img.mul(255).add(0.5).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def resize_size(h, w, size):
    """torchvision ``Resize(int)`` output size rule (test.py:16-24 via transforms.Resize):
    shorter side -> size, longer side -> int(size * long / short); unchanged if already equal.
    PARITY UNPINNED by import; known answers: 933x700 (WxH) at 256 -> 341x256."""
    if size == 0:
        return h, w
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def pil_resample_taps(in_size, out_size):
    """Pillow libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter over
    the whole axis (box = (0, in_size)): per output index (first source index, fixed-point taps).  Pillow is a third-party
    dependency of the reference (requirements.txt:1, unpinned; torchvision 0.13.1's Resize on a PIL image calls
    ``Image.resize(size, BILINEAR)``, test.py:16-24); PINNED: tests/test_oracle_golden.py checks this restatement against the
    Pillow installed in the image (12.2.0) on the callers' size pairs and random ones, byte for byte."""
    import numpy as np

    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    taps = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        t = [abs(((x + xmin) - center + 0.5) * ss) for x in range(xmax)]
        w = [1.0 - v if v < 1.0 else 0.0 for v in t]        # bilinear_filter
        ww = sum(w, 0.0)                                     # left-to-right double additions, as the C loop does
        if ww != 0.0:
            w = [v / ww for v in w]
        taps.append((xmin, np.array([int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in w], dtype=np.int64)))
    return taps


def resize_pil_bilinear_u8(img, size):
    """``PIL.Image.resize(size, BILINEAR)`` of a uint8 [H,W,C] array, restated (Resample.c ImagingResample, 8 bits per channel):
    horizontal pass into a uint8 intermediate, vertical pass, each ``clip8((2^21 + sum in * k) >> 22)``.  ``size`` = (width,
    height).  See ``pil_resample_taps`` for the pin."""
    import numpy as np

    def one_axis(a, taps):                                    # resamples axis 1 of [R, S, C]
        out = np.empty((a.shape[0], len(taps), a.shape[2]), dtype=np.uint8)
        src = a.astype(np.int64)
        for xx, (xmin, k) in enumerate(taps):
            acc = (1 << 21) + (src[:, xmin:xmin + len(k), :] * k[None, :, None]).sum(axis=1)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255)
        return out

    wo, ho = size
    tmp = one_axis(img, pil_resample_taps(img.shape[1], wo))
    return one_axis(tmp.transpose(1, 0, 2), pil_resample_taps(img.shape[0], ho)).transpose(1, 0, 2)


def psnr(a, b):
    """Style_3DGS/utils/image_utils.py:17-19 — 20*log10(1/sqrt(mse)) per image on [0,1] data."""
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def warp_u8(prev, flow):
    """Reference video/utils.py:89-105 (warp_image): cv2.remap(uint8 HWC, x + flow[0], y + flow[1], INTER_LINEAR,
    BORDER_REFLECT).  cv2 is a third-party dependency absent from this image (requirements.txt:6 opencv-contrib-python,
    version not pinned); this restates OpenCV 4.x's published fixed-point path (modules/imgproc/src/imgwarp.cpp:
    RemapInvoker map conversion + remapBilinear<FixedPtCast<int, uchar, 15>>, INTER_BITS = 5): map rounded half-to-even
    to 1/32 px, bilinear weights from the 32x32 table scaled to 2^15, (sum + 2^14) >> 15, borderInterpolate(REFLECT).
    PARITY UNPINNED against OpenCV itself (no cv2 here to generate vectors)."""
    import numpy as np

    h, w, c = prev.shape
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    mx = (x + flow[0]).astype(np.float32)                  # video/utils.py:98-99
    my = (y + flow[1]).astype(np.float32)
    ix = np.rint(mx * np.float32(32)).astype(np.int64)     # cvRound(sX * INTER_TAB_SIZE)
    iy = np.rint(my * np.float32(32)).astype(np.int64)
    sx, sy = np.clip(ix >> 5, -32768, 32767), np.clip(iy >> 5, -32768, 32767)   # saturate_cast<short>
    fx, fy = (ix & 31)[..., None], (iy & 31)[..., None]

    def refl(v, n):                                        # borderInterpolate(BORDER_REFLECT): fedcba|abcdefgh|hgfedcb
        v = np.mod(v, 2 * n)
        return np.where(v < n, v, 2 * n - 1 - v)

    x0, x1, y0, y1 = refl(sx, w), refl(sx + 1, w), refl(sy, h), refl(sy + 1, h)
    p = prev.astype(np.int64)
    acc = (p[y0, x0] * ((32 - fx) * (32 - fy) * 32) + p[y0, x1] * (fx * (32 - fy) * 32)
           + p[y1, x0] * ((32 - fx) * fy * 32) + p[y1, x1] * (fx * fy * 32) + (1 << 14)) >> 15
    return acc.astype(np.uint8)


def warp_blend_u8(cur, prev, flow, alpha):
    """warp_u8 followed by reference video/utils.py:223-229 (blend_images, numpy float32 semantics) on uint8 HWC frames;
    flow [2,H,W]."""
    import numpy as np

    warped = warp_u8(prev, flow).astype(np.float32) / np.float32(255.0)
    blended = np.float32(alpha) * (cur.astype(np.float32) / np.float32(255.0)) + np.float32(1 - alpha) * warped
    return np.clip(blended * np.float32(255), 0, 255).astype(np.uint8)


def temporal_blend(frames, flows, alpha):
    """The frame-to-frame recurrence of video/utils.py:352-369 on already stylised uint8 frames [n,H,W,C]:
    frame 0 is kept; frame i = blend(frame_i, warp(result_{i-1}, flow_{i-1}))."""
    import numpy as np

    out = [frames[0]]
    for i in range(1, len(frames)):
        out.append(warp_blend_u8(frames[i], out[-1], flows[i - 1], alpha))
    return np.stack(out)


def _area_tab(ssize, dsize, scale):
    """OpenCV 4.x modules/imgproc/src/resize.cpp computeResizeAreaTab (double arithmetic, float weights): for every output
    index the list of (source index, weight) taps — a partial first cell, whole cells, a partial last cell."""
    import math

    import numpy as np

    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        taps = []
        if sx1 - fsx1 > 1e-3:
            taps.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            taps.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            taps.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(taps)
    return tab


def _area_linear_coeffs(ssize, dsize, clamp_last):
    """cv::resize's coefficient loop for INTER_AREA outside the true-area branch (resize.cpp, `area_mode`): per output index
    the left source index and the two 11-bit fixed-point weights.  sx = cvFloor(d * scale); f = (float)((d + 1) - (sx + 1) *
    inv_scale), f <= 0 -> 0 else f - cvFloor(f); along x an index at the last source column gets f = 0 (the xmax rule);
    weights = saturate_cast<short>((1.f - f, f) * 2048) = round-half-even."""
    import math

    import numpy as np

    inv_scale = dsize / ssize
    scale = 1.0 / inv_scale
    idx, w0, w1 = [], [], []
    for d in range(dsize):
        sx = math.floor(d * scale)
        f = np.float32((d + 1) - (sx + 1) * inv_scale)
        f = np.float32(0) if f <= 0 else np.float32(f - np.float32(math.floor(f)))
        if clamp_last and sx >= ssize - 1:
            f, sx = np.float32(0), ssize - 1
        idx.append(sx)
        w0.append(int(np.rint((np.float32(1) - f) * np.float32(2048))))
        w1.append(int(np.rint(f * np.float32(2048))))
    return np.array(idx), np.array(w0, dtype=np.int64), np.array(w1, dtype=np.int64)


def _resize_area_linear_u8(img, wo, ho):
    """INTER_AREA with an enlarged axis = cv::resize's generic linear path with area-mode coefficients on uint8:
    HResizeLinear (int row buffer = S[sx] * a0 + S[sx + 1] * a1, 2048 = one), source rows sy and min(sy + 1, H - 1),
    VResizeLinear's fixed-point combine ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    import numpy as np

    hi, wi, c = img.shape
    sx, a0, a1 = _area_linear_coeffs(wi, wo, True)
    sy, b0, b1 = _area_linear_coeffs(hi, ho, False)
    S = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, wi - 1)
    rows = S[:, sx, :] * a0[None, :, None] + S[:, sx1, :] * a1[None, :, None]            # [hi, wo, c]
    r0, r1 = np.minimum(sy, hi - 1), np.minimum(sy + 1, hi - 1)
    out = (((b0[:, None, None] * (rows[r0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[r1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def resize_area_u8(src, dsize):
    """cv2.resize(src, dsize, interpolation=cv2.INTER_AREA) for uint8 HWC (or HW) images, ``dsize`` = (width, height), as the
    video caller applies it to every stylised frame (reference video/utils.py:352-353).  cv2 is a third-party dependency
    absent from this image (requirements.txt:6 opencv-contrib-python, unpinned); this restates the published true-area branch
    of OpenCV 4.x's cv::resize (modules/imgproc/src/resize.cpp): same size -> copy; integer scales -> resizeAreaFast_
    (int box sum, saturate_cast<uchar>(sum * (1.f / area)) = round-half-even; (a+b+c+d+2)>>2 for 2x2 with 1/3/4 channels);
    other scales -> resizeArea_<uchar, float> (computeResizeAreaTab taps, float accumulation over x taps then rows, in
    OpenCV's order, round-half-even).  When an axis is ENLARGED (scale < 1 on either axis) cv::resize leaves the true-area
    branch and emulates INTER_AREA with its 11-bit fixed-point bilinear pass in "area mode" (_resize_area_linear_u8 below).
    PARITY UNPINNED against OpenCV itself (no cv2 here to generate vectors); known answers in tests/test_oracle_golden.py."""
    import numpy as np

    img = np.asarray(src)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    hi, wi, c = img.shape
    wo, ho = int(dsize[0]), int(dsize[1])
    if (ho, wo) == (hi, wi):
        out = img.copy()
        return out[:, :, 0] if squeeze else out
    if ho > hi or wo > wi:
        out = _resize_area_linear_u8(img, wo, ho)
        return out[:, :, 0] if squeeze else out
    scale_x, scale_y = 1.0 / (wo / wi), 1.0 / (ho / hi)          # cv::resize: scale = 1. / inv_scale, inv_scale = dsize / ssize
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))     # saturate_cast<int>(double) = cvRound
    eps = np.finfo(np.float64).eps
    if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps:    # is_area_fast
        blocks = img[: ho * isy, : wo * isx].reshape(ho, isy, wo, isx, c).astype(np.int64).sum(axis=(1, 3))
        if isx == 2 and isy == 2 and c in (1, 3, 4):
            out = ((blocks + 2) >> 2).astype(np.uint8)
        else:
            v = blocks.astype(np.float32) * (np.float32(1.0) / np.float32(isx * isy))
            out = np.clip(np.rint(v), 0, 255).astype(np.uint8)
        return out[:, :, 0] if squeeze else out
    xtab, ytab = _area_tab(wi, wo, scale_x), _area_tab(hi, ho, scale_y)
    S = img.astype(np.float32)
    # row buffers: buf[sy][dx] = sum over the x taps of dx, in tap order, float32
    buf = np.zeros((hi, wo, c), dtype=np.float32)
    for dx, taps in enumerate(xtab):
        acc = np.zeros((hi, c), dtype=np.float32)
        for sx, a in taps:
            acc = acc + S[:, sx, :] * a
        buf[:, dx, :] = acc
    out = np.zeros((ho, wo, c), dtype=np.uint8)
    for dy, taps in enumerate(ytab):
        acc = None
        for sy, b in taps:
            acc = b * buf[sy] if acc is None else acc + b * buf[sy]
        out[dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out
