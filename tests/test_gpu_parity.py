"""GPU parity tests: the HIP path (through the C ABI via ctypes) against the CPU oracle and the golden
fixtures produced by the reference's own code.  Run on the MI355X box with ``-m gpu``.

Tolerances (fp32 path, values O(1)-O(10)): per-layer / feature tensors rtol 2e-4 + atol 2e-4 (summation
order differs from oneDNN over K up to 4608); end-to-end images additionally PSNR >= 60 dB after the
reference's own [0,1] clamp (target in BASELINE.json is >= 40 dB); uint8 quantisation bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import applied_image_processing_amd.synth as synth
from conftest import golden
from oracle import adain_oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-4


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import applied_image_processing_amd.runtime as rt

    rt.lib()  # fails loudly if the .so is missing
    return rt


@pytest.fixture(scope="module")
def nets(weights):
    from applied_image_processing_amd.AdaIN import net

    vgg_sd, dec_sd = weights
    full = synth.to_torch(synth.vgg_state_dict(0, full=True))
    net.vgg.load_state_dict(full)
    net.decoder.load_state_dict(dec_sd)
    net.vgg.to("cuda").eval()
    net.decoder.to("cuda").eval()
    return net.vgg, net.decoder


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(gpu, ref, rtol=RTOL, atol=ATOL):
    g = gpu.detach().cpu().contiguous().numpy() if torch.is_tensor(gpu) else gpu
    r = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    np.testing.assert_allclose(g, r, rtol=rtol, atol=atol)


def psnr01(a, b):
    a, b = a.detach().cpu().clamp(0, 1), b.detach().cpu().clamp(0, 1)
    return float(O.psnr(a, b).min())


# ---- single conv layers (F(4,3) x F(2,3), the one form the library holds), gather modes, ragged tiles -----------------------
def test_conv3x3_f43xf23_random_layers(rt):
    """The product's 3x3 kernel (F(4,3) x F(2,3), form 5) on random layers: channel counts from 16 to 512, sizes from 2 x 2 to
    150 x 200 (one-tile and persistent launches, ragged tiles), batches, direct / up-sampled source, output pool, ReLU on / off."""
    rng = np.random.default_rng(31)
    for case in range(36):
        cin, cout = int(rng.choice([16, 32, 48, 64, 128, 256, 512])), int(rng.choice([32, 64, 96, 128, 256, 512]))
        n, hs, ws = int(rng.integers(1, 4)), int(rng.integers(2, 150)), int(rng.integers(2, 200))
        if cin * cout >= 128 * 256:
            hs, ws = min(hs, 70), min(ws, 90)                                    # keeps the CPU reference quick
        up, relu = bool(case % 3 == 1), bool(rng.random() < 0.7)
        if up:
            hs, ws = max(1, hs // 2), max(1, ws // 2)
        pool = bool(case % 3 == 2)
        x = T(synth.uniform_sym(2000 + case, (n, cin, hs, ws), 1.0))
        w = T(synth.uniform_sym(2100 + case, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
        b = T(synth.uniform_sym(2200 + case, (cout,), 0.1))
        src = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
        if min(src.shape[2:]) < 2:
            continue
        ref = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="reflect"), w, b)
        if relu:
            ref = F.relu(ref)
        if pool:
            ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
        out = rt.conv3x3_wino(x.cuda().permute(0, 2, 3, 1).contiguous(), rt.conv3x3_wino_pack(w.cuda(), 5), b.cuda(), cout,
                              rt.SRC_UP2X if up else rt.SRC_DIRECT, relu, pool, 5)
        assert tuple(out.shape) == (n,) + tuple(ref.shape[2:]) + (cout,), (case, cin, cout, n, hs, ws, up, pool)
        np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL,
                                   err_msg=f"case {case}: {cin}->{cout} n={n} {hs}x{ws} up={up} pool={pool} relu={relu}")


@pytest.mark.parametrize("mode", ["direct", "up"])
@pytest.mark.parametrize("shape", [(1, 128, 128, 16, 32), (2, 128, 64, 9, 37), (1, 256, 128, 21, 70), (1, 512, 256, 5, 6), (1, 64, 64, 8, 8)])
def test_conv3x3_winograd_vs_oracle(rt, mode, shape):
    """The F(4,3) x F(2,3) kernel on small maps (one-tile launches) against torch's direct convolution: ReLU on / off, fused pool."""
    m_tiles = 5
    n, cin, cout, hs, ws = shape
    x = T(synth.uniform_sym(400 + cin, (n, cin, hs, ws), 1.0))
    w = T(synth.uniform_sym(500 + cout, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
    b = T(synth.uniform_sym(600 + cout, (cout,), 0.1))
    src = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    pre = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="reflect"), w, b)
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    packed = rt.conv3x3_wino_pack(w.cuda(), m_tiles)
    m = rt.SRC_UP2X if mode == "up" else rt.SRC_DIRECT
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=True, m_tiles=m_tiles).permute(0, 3, 1, 2), F.relu(pre))
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=m_tiles).permute(0, 3, 1, 2), pre)
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=True, pool_out=True, m_tiles=m_tiles).permute(0, 3, 1, 2),
          F.max_pool2d(F.relu(pre), 2, 2, 0, ceil_mode=True))


@pytest.mark.parametrize("mode,shape", [("direct", (1, 64, 128, 250, 203)), ("direct", (2, 32, 64, 131, 257)), ("up", (1, 64, 64, 121, 150)),
                                        ("direct", (3, 128, 32, 6, 40)), ("direct", (1, 256, 256, 64, 96))])
def test_conv3x3_winograd_f43_full_grids(rt, mode, shape):
    """The F(4,3) x F(2,3) form on grids that fill every CU with two workgroups (co-resident waves queueing LDS and VMEM
    work), ragged edges (H, W not multiples of the 8 x 32 tile), several images and channel tiles; direct, ReLU, pooled."""
    n, cin, cout, hs, ws = shape
    x = T(synth.uniform_sym(420 + cin, (n, cin, hs, ws), 1.0))
    w = T(synth.uniform_sym(520 + cout, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
    b = T(synth.uniform_sym(620 + cout, (cout,), 0.1))
    src = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    pre = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="reflect"), w, b)
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    packed = rt.conv3x3_wino_pack(w.cuda(), 5)
    m = rt.SRC_UP2X if mode == "up" else rt.SRC_DIRECT
    out = rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5)
    close(out.permute(0, 3, 1, 2), pre)
    assert torch.equal(out, rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5))      # deterministic
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=True, pool_out=True, m_tiles=5).permute(0, 3, 1, 2),
          F.max_pool2d(F.relu(pre), 2, 2, 0, ceil_mode=True))


@pytest.mark.parametrize("mode,shape", [("direct", (1, 64, 128, 300, 400)), ("direct", (4, 32, 64, 150, 200)), ("direct", (1, 128, 256, 150, 200)),
                                        ("up", (1, 64, 64, 75, 100)), ("direct", (3, 48, 64, 150, 200)), ("direct", (2, 64, 32, 16, 16)),
                                        ("up", (2, 64, 128, 150, 200)), ("direct", (1, 64, 64, 57, 112))])
def test_conv3x3_winograd_f43_16x16_tile_geometry(rt, mode, shape):
    """Shapes whose feature maps the 16 x 16 tile geometry covers with fewer tiles than the default 8 x 32 (W = 400 / 200 / 114 ...):
    the launcher then runs the second layout of the kernel (4 x 8 Winograd tiles, 18 x 18 halo) - persistent and one-tile
    launches, two- / three- / many-stage K loops, batches, up-sampled source, ragged edges, fused pool, determinism."""
    import applied_image_processing_amd.arch as arch

    n, cin, cout, hs, ws = shape
    h, w = (2 * hs, 2 * ws) if mode == "up" else (hs, ws)
    assert arch.wino4_geometry([(n, h, w)]) == 1
    x = T(synth.uniform_sym(440 + cin, (n, cin, hs, ws), 1.0))
    wt = T(synth.uniform_sym(540 + cout, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
    b = T(synth.uniform_sym(640 + cout, (cout,), 0.1))
    src = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    pre = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="reflect"), wt, b)
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    packed = rt.conv3x3_wino_pack(wt.cuda(), 5)
    m = rt.SRC_UP2X if mode == "up" else rt.SRC_DIRECT
    out = rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5)
    close(out.permute(0, 3, 1, 2), pre)
    assert torch.equal(out, rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5))      # deterministic
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=True, pool_out=True, m_tiles=5).permute(0, 3, 1, 2),
          F.max_pool2d(F.relu(pre), 2, 2, 0, ceil_mode=True))


@pytest.mark.parametrize("mode,shape", [("direct", (2, 32, 64, 256, 250)), ("direct", (1, 64, 128, 296, 250)), ("up", (1, 64, 64, 200, 176)),
                                        ("direct", (3, 48, 64, 203, 340))])
def test_conv3x3_winograd_f43_persistent_tile_lists(rt, mode, shape):
    """Launches with at least two tiles per resident workgroup take the persistent form of the F(4,3) x F(2,3) kernel: tile
    hand-over (prefetched halo, weight-ring wrap into the next tile's channel tile, LDS reuse), the two-stage special case
    (cin = 32), three-stage (cin = 48), several images, ragged edges, fused pool."""
    n, cin, cout, hs, ws = shape
    h, w = (2 * hs, 2 * ws) if mode == "up" else (hs, ws)
    import applied_image_processing_amd.arch as arch

    assert arch.wino4_geometry([(n, h, w)]) == 0                                   # the default 8 x 32 geometry
    assert ((w + 31) // 32) * ((h + 7) // 8) * (cout // 32) * n >= 1024          # the launcher's persistence threshold on 256 CUs
    x = T(synth.uniform_sym(430 + cin, (n, cin, hs, ws), 1.0))
    wt = T(synth.uniform_sym(530 + cout, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
    b = T(synth.uniform_sym(630 + cout, (cout,), 0.1))
    src = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    pre = F.conv2d(F.pad(src, (1, 1, 1, 1), mode="reflect"), wt, b)
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    packed = rt.conv3x3_wino_pack(wt.cuda(), 5)
    m = rt.SRC_UP2X if mode == "up" else rt.SRC_DIRECT
    out = rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5)
    close(out.permute(0, 3, 1, 2), pre)
    assert torch.equal(out, rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=False, m_tiles=5))      # deterministic
    close(rt.conv3x3_wino(xg, packed, b.cuda(), cout, m, relu=True, pool_out=True, m_tiles=5).permute(0, 3, 1, 2),
          F.max_pool2d(F.relu(pre), 2, 2, 0, ceil_mode=True))


def test_conv3x3_rejects_bad_shapes(rt):
    x = torch.zeros(1, 4, 4, 24, device="cuda")             # cin not a multiple of 16
    with pytest.raises(rt.AdainHipError):
        rt.conv3x3_wino(x, torch.zeros(24 * 64 * 24, device="cuda"), torch.zeros(64, device="cuda"), 64)
    x = torch.zeros(1, 1, 8, 64, device="cuda")  # H == 1: reflection pad impossible
    with pytest.raises(rt.AdainHipError):
        rt.conv3x3_wino(x, torch.zeros(64 * 64 * 24, device="cuda"), torch.zeros(64, device="cuda"), 64)
    x = torch.zeros(1, 8, 8, 64, device="cuda")
    with pytest.raises(rt.AdainHipError, match="retired"):          # the F(2x2,3x3) forms went with the diagnostic families (round 6)
        rt.conv3x3_wino(x, torch.zeros(64 * 64 * 24, device="cuda"), torch.zeros(64, device="cuda"), 64, m_tiles=3)
    with pytest.raises(rt.AdainHipError, match="retired"):
        rt.conv3x3_wino_pack(torch.zeros(64, 64, 3, 3, device="cuda"), 3)
    with pytest.raises(rt.AdainHipError):
        rt.encode(torch.zeros(1, 3, 8, 64), torch.zeros(8))  # CPU tensor: no fallback


# ---- encoder / decoder / full path against golden fixtures from the reference --------------------------
def test_case_a_golden(rt, nets, weights):
    vgg, dec = nets
    g = golden("case_a.npz")
    c, s = T(synth.image(11, 1, 64, 64)).cuda(), T(synth.image(12, 1, 48, 80)).cuda()
    from applied_image_processing_amd.AdaIN import function as fn, test as t

    cf, sf = vgg(c), vgg(s)
    assert tuple(cf.shape) == (1, 512, 8, 8)
    close(cf, g["content_f"])
    close(sf, g["style_f"])
    m, sd = fn.calc_mean_std(cf)
    assert tuple(m.shape) == (1, 512, 1, 1)
    close(m, g["mean"], 1e-5, 1e-5)
    close(sd, g["std"], 1e-5, 1e-5)
    # NCHW-contiguous input takes the other kernel
    m2, sd2 = fn.calc_mean_std(T(g["content_f"]).cuda())
    close(m2, g["mean"], 1e-5, 1e-6)
    close(sd2, g["std"], 1e-5, 1e-6)
    close(fn.adaptive_instance_normalization(T(g["content_f"]).cuda(), T(g["style_f"]).cuda()), g["adain"], 1e-4, 1e-4)
    close(fn.adaptive_instance_normalization(cf, sf), g["adain"])
    for alpha, key in ((0.5, "out_a05"), (1.0, "out_a10")):
        out = t.style_transfer_simple(vgg, dec, c, s, alpha)
        assert tuple(out.shape) == (1, 3, 64, 64)
        close(out, g[key], 5e-4, 5e-4)
        assert psnr01(out, T(g[key])) >= 60.0


def test_case_f_config1_reference_sample_images(rt, nets):
    """BASELINE config 1 on the GPU: the reference's own sample images (resized to 256; uint8 data in the fixture) through
    style_transfer_simple - from the float tensors as run_depth.py / adain_inference build them, and from the uint8 frames with
    ToTensor inside the first layer's kernel - against the reference's output."""
    vgg, dec = nets
    g = golden("case_f.npz")
    from applied_image_processing_amd.AdaIN import test as t

    cu8 = T(g["content_u8"]).cuda()
    # ToTensor on the HOST, as the reference runs it (a correctly rounded fp32 division; torch's GPU `div` multiplies by 1/255)
    c = T(g["content_u8"]).permute(2, 0, 1).float().div(255).unsqueeze(0).contiguous().cuda()
    s = T(g["style_u8"]).permute(2, 0, 1).float().div(255).unsqueeze(0).contiguous().cuda()
    out = t.style_transfer_simple(vgg, dec, c, s, 0.5)
    assert tuple(out.shape) == (1, 3, 256, 256)
    close(out, g["out"], 5e-4, 5e-4)
    assert psnr01(out, T(g["out"])) >= 60.0
    rel = float((out.cpu() - T(g["out"])).norm() / T(g["out"]).norm())
    assert rel < 1e-4, rel
    # the same forward from the decoded frames (what the job drivers upload)
    from applied_image_processing_amd.engine import AdaINEngine

    eng = AdaINEngine(vgg.state_dict(), dec.state_dict(), "cuda:0")
    eng.set_style(s)
    assert torch.equal(eng.stylize(cu8.unsqueeze(0), 0.5), out)


def test_case_b_odd_depth_mask_golden(rt, nets):
    vgg, dec = nets
    g = golden("case_b.npz")
    from applied_image_processing_amd.AdaIN import test as t

    c, s = T(synth.image(21, 1, 45, 67)).cuda(), T(synth.image(22, 1, 50, 38)).cuda()
    depth = T(synth.smooth_depth(23, 90, 134)).cuda()
    cf = vgg(c)
    assert tuple(cf.shape) == (1, 512, 6, 9)
    close(cf, g["content_f"])
    close(t.compute_stylization_strength_map(depth, (6, 9), 0.15, 20), g["pmap"], 1e-4, 1e-5)
    close(t.compute_stylization_strength_map(depth, (11, 7), 0.4, 7.5), g["pmap_other"], 1e-4, 1e-5)
    pz = t.compute_stylization_strength_map(torch.zeros(20, 30), (6, 9), 0.15, 20)
    assert float(pz.abs().max()) == 0.0 and tuple(pz.shape) == (1, 1, 6, 9)
    out = t.style_transfer_simple(vgg, dec, c, s, 0.5)
    assert tuple(out.shape) == (1, 3, 48, 72)
    close(out, g["out_simple"], 5e-4, 5e-4)
    outd = t.style_transfer(vgg, dec, c, s, depth, 1.0, 0.15, 20)
    close(outd, g["out_depth"], 5e-4, 5e-4)
    assert psnr01(outd, T(g["out_depth"])) >= 60.0
    s4 = torch.cat([s, T(synth.image(24, 1, 50, 38, c=1)).cuda()], dim=1)
    close(t.style_transfer(vgg, dec, c, s4, depth, 1.0, 0.15, 20), g["out_depth_rgba"], 5e-4, 5e-4)
    mask3 = (c[0].cpu().numpy() > 0.3)
    mask1 = (synth.image(25, 1, 30, 40, c=1)[0] > 0.5).astype(np.uint8)
    gs = T(g["out_simple"]).cuda()
    close(t.composite_with_mask(c, gs, mask3), g["comp3"], 1e-5, 1e-5)
    close(t.composite_with_mask(c, gs, mask1), g["comp1"], 1e-5, 1e-5)
    with pytest.raises(AssertionError):
        t.style_transfer_simple(vgg, dec, c, s, 1.5)
    with pytest.raises(AssertionError):
        t.style_transfer(vgg, dec, c, s, depth, 1.0, 1.5, 20)


def test_case_c_batch_golden(rt, nets):
    vgg, dec = nets
    g = golden("case_c.npz")
    from applied_image_processing_amd.AdaIN import function as fn, test as t

    c, s = T(synth.image(31, 2, 40, 56)).cuda(), T(synth.image(32, 2, 33, 47)).cuda()
    close(vgg(c), g["content_f"])
    m, sd = fn.calc_mean_std(vgg(s))
    close(m, g["style_mean"], 1e-5, 1e-5)
    close(sd, g["style_std"], 1e-5, 1e-5)
    close(t.style_transfer_simple(vgg, dec, c, s, 0.7), g["out_a07"], 5e-4, 5e-4)
    with pytest.raises(AssertionError):
        fn.adaptive_instance_normalization(vgg(c), vgg(s[:1]))          # (N, C) mismatch (function.py:16)
    with pytest.raises(AssertionError):
        fn.calc_mean_std(torch.zeros(4, 4, device="cuda"))              # not 4-D (function.py:7)


# ---- pixel kernels vs torch CPU ----------------------------------------------------------------------
def test_pixel_kernels_vs_oracle(rt):
    x = T(synth.uniform_sym(41, (2, 3, 37, 53), 2.0))
    for size in ((64, 80), (20, 31), (37, 53), (1, 1)):
        close(rt.resize_bilinear(x.cuda(), size), F.interpolate(x, size=size, mode="bilinear", align_corners=False), 1e-5, 1e-5)
        g = rt.resize_nearest(x.cuda(), size).cpu()
        assert torch.equal(g, F.interpolate(x, size=size, mode="nearest"))
    img = T(synth.uniform_sym(42, (2, 3, 19, 23), 0.8)) + 0.5      # spans < 0 and > 1
    q = rt.quantize_u8(img.cuda()).cpu()
    assert torch.equal(q, O.quantize_u8(img))                     # bit-exact
    f = T(synth.uniform_sym(43, (2, 8, 5, 7), 1.0))
    assert torch.equal(rt.nchw_to_nhwc(f.cuda()).cpu(), f.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(rt.nhwc_to_nchw(rt.nchw_to_nhwc(f.cuda())).cpu(), f)


def test_pixel_kernels_random_sizes(rt):
    """Random plane counts and sizes (odd widths: every vector tail) through the pixel kernels of the mask composite, the
    strength map and the quantiser."""
    rng = np.random.default_rng(23)
    for case in range(40):
        n, c = int(rng.integers(1, 4)), int(rng.choice([1, 3, 3, 4]))
        hi, wi, ho, wo = (int(v) for v in rng.integers(1, 150, 4))
        x = T(synth.uniform_sym(300 + case, (n, c, hi, wi), 2.0))
        # one ulp of a source coordinate near 150 (fused vs separate multiply-subtract) times the local gradient: a few 1e-5
        close(rt.resize_bilinear(x.cuda(), (ho, wo)), F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False), 1e-5, 5e-5)
        assert torch.equal(rt.resize_nearest(x.cuda(), (ho, wo)).cpu(), F.interpolate(x, size=(ho, wo), mode="nearest"))
        img = T(synth.uniform_sym(400 + case, (n, c, ho, wo), 0.8)) + 0.5
        assert torch.equal(rt.quantize_u8(img.cuda()).cpu(), O.quantize_u8(img))
        if c == 3:
            content, sty = T(synth.image(500 + case, n, ho, wo)), T(synth.uniform_sym(600 + case, (n, 3, ho, wo), 1.0))
            mc, mn = int(rng.choice([1, 3])), int(rng.choice([1, n]))
            mask = T((synth.image(700 + case, mn, ho, wo, c=mc) > 0.5).astype(np.float32))
            want = content * (1 - mask) + sty * mask
            close(rt.mask_composite(content.cuda(), sty.cuda(), mask.cuda()), want, 1e-6, 1e-6)
        h0, w0, hc, wc = int(rng.integers(2, 300)), int(rng.integers(2, 300)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        d = T(synth.smooth_depth(800 + case, h0, w0))
        off, prom = float(rng.random() * 0.9), float(rng.random() * 30)
        close(rt.strength_map(d.cuda(), hc, wc, off, prom), O.compute_stylization_strength_map(d, (hc, wc), off, prom), 1e-4, 2e-5)


def test_video_warp_blend_vs_oracle(rt):
    h, w = 45, 61
    cur = (synth.image(81, 1, h, w)[0].transpose(1, 2, 0) * 255).astype(np.uint8)
    prev = (synth.image(82, 1, h, w)[0].transpose(1, 2, 0) * 255).astype(np.uint8)
    flow = synth.uniform_sym(83, (2, h, w), 6.0)           # up to 6 px, leaves the frame near the borders
    out = rt.warp_blend_u8(T(cur).cuda(), T(prev).cuda(), T(flow).cuda(), 0.7).cpu().numpy()
    ref = O.warp_blend_u8(cur, prev, flow, 0.7)
    assert np.array_equal(out, ref)                        # fixed-point warp + uncontracted fp32 blend: bit-exact
    # every uint8 value through both divisions by 255 (the kernel divides by a corrected reciprocal: must equal numpy's float32 /),
    # several alphas, with a 4-pixel-aligned frame (vector kernel) and an odd one (scalar kernel)
    for hh, ww in ((16, 64), (15, 67)):
        ramp = (np.arange(hh * ww * 3) % 256).astype(np.uint8).reshape(hh, ww, 3)
        other = ((np.arange(hh * ww * 3) * 7 + 3) % 256).astype(np.uint8).reshape(hh, ww, 3)
        fl = synth.uniform_sym(84, (2, hh, ww), 1.5)
        for al in (0.7, 0.5, 0.123, 1.0, 0.0):
            got = rt.warp_blend_u8(T(ramp).cuda(), T(other).cuda(), T(fl).cuda(), al).cpu().numpy()
            assert np.array_equal(got, O.warp_blend_u8(ramp, other, fl, al)), (hh, ww, al)
    # displacements of several frame sizes (the out-of-line reflection path)
    big = synth.uniform_sym(85, (2, h, w), 400.0)
    assert np.array_equal(rt.warp_blend_u8(T(cur).cuda(), T(prev).cuda(), T(big).cuda(), 0.7).cpu().numpy(), O.warp_blend_u8(cur, prev, big, 0.7))
    # zero flow, alpha = 1 reproduces the current frame exactly; alpha = 0 the previous one
    z = torch.zeros(2, h, w).cuda()
    assert np.array_equal(rt.warp_blend_u8(T(cur).cuda(), T(prev).cuda(), z, 1.0).cpu().numpy(), cur)
    assert np.abs(rt.warp_blend_u8(T(cur).cuda(), T(prev).cuda(), z, 0.0).cpu().numpy().astype(int) - prev.astype(int)).max() <= 1


def test_video_temporal_recurrence_vs_oracle(rt):
    from applied_image_processing_amd import engine

    n, h, w = 5, 40, 56
    frames = np.stack([(synth.image(90 + i, 1, h, w)[0].transpose(1, 2, 0) * 255).astype(np.uint8) for i in range(n)])
    flows = np.stack([synth.uniform_sym(95 + i, (2, h, w), 3.0) for i in range(n - 1)])
    out = engine.temporal_blend(T(frames).cuda(), T(flows).cuda(), 0.7).cpu().numpy()
    assert np.array_equal(out, O.temporal_blend(frames, flows, 0.7))
    assert np.array_equal(out[0], frames[0])
    one = engine.temporal_blend(T(frames[:1]).cuda(), T(flows[:0]).cuda(), 0.7)        # a single frame passes through
    assert np.array_equal(one.cpu().numpy(), frames[:1])
    with pytest.raises(rt.AdainHipError):
        engine.temporal_blend(T(frames).cuda(), T(flows[:2]).cuda(), 0.7)


def test_mean_std_shapes_and_precision(rt):
    from applied_image_processing_amd.AdaIN import function as fn

    # large mean / small variance: fp64 accumulation must not cancel
    x = T(synth.uniform_sym(51, (2, 512, 33, 29), 0.01)) + 100.0
    rm, rs = O.calc_mean_std(x.double())
    m, s = fn.calc_mean_std(x.cuda())
    close(m, rm.float(), 1e-6, 0)
    close(s, rs.float(), 1e-3, 1e-6)
    m, s = fn.calc_mean_std(x.cuda().contiguous(memory_format=torch.channels_last))
    close(m, rm.float(), 1e-6, 0)
    close(s, rs.float(), 1e-3, 1e-6)
    # small channel count, NCHW
    y = T(synth.uniform_sym(52, (1, 3, 8, 8), 1.0))
    rm, rs = O.calc_mean_std(y)
    m, s = fn.calc_mean_std(y.cuda())
    close(m, rm, 1e-5, 1e-6)
    close(s, rs, 1e-5, 1e-6)


# ---- full-size config 2: 1024x1024 content, 512x512 style (BASELINE.json configs[1]) ------------------
def test_config2_1024_vs_oracle_and_properties(rt, nets, weights):
    vgg, dec = nets
    vgg_sd, dec_sd = weights
    from applied_image_processing_amd.AdaIN import test as t

    c, s = T(synth.image(3, 1, 1024, 1024)), T(synth.image(4, 1, 512, 512))
    out = t.style_transfer_simple(vgg, dec, c.cuda(), s.cuda(), 0.5)
    assert tuple(out.shape) == (1, 3, 1024, 1024)
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5)
    rel = float((out.cpu() - ref).norm() / ref.norm())
    p = psnr01(out, ref)
    print(f"config2: relative L2 {rel:.3e}, PSNR {p:.1f} dB")
    assert rel < 1e-4 and p >= 60.0
    # idempotent / deterministic: the same call twice is bitwise identical
    out2 = t.style_transfer_simple(vgg, dec, c.cuda(), s.cuda(), 0.5)
    assert torch.equal(out, out2)
    # batching: two copies in a batch give the single-image result for both
    cb = torch.cat([c, c]).cuda()
    ob = dec(t._adain_blend(vgg(cb), vgg(torch.cat([s, s]).cuda()), alpha=0.5))
    assert torch.equal(ob[0], out[0]) and torch.equal(ob[1], out[0])
    # alpha = 0 reduces to decoder(encoder(content)) regardless of the style
    o0 = t.style_transfer_simple(vgg, dec, c.cuda(), s.cuda(), 0.0)
    o0b = dec(vgg(c.cuda()))
    close(o0, o0b, 1e-6, 1e-6)


def test_adain_inference_end_to_end(rt, tmp_path, weights):
    """Host wrapper: PIL in, JPEG path out, with the weights loaded from .pth files in the reference's
    state_dict layout."""
    from PIL import Image
    from applied_image_processing_amd.AdaIN import test as t
    from applied_image_processing_amd.AdaIN import adain_inference, get_style_embeddings

    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(synth.to_torch(synth.decoder_state_dict(0)), tmp_path / "dec.pth")
    cimg = Image.fromarray((synth.image(61, 1, 96, 120)[0].transpose(1, 2, 0) * 255).astype(np.uint8))
    simg = Image.fromarray((synth.image(62, 1, 70, 90)[0].transpose(1, 2, 0) * 255).astype(np.uint8))
    p = adain_inference(cimg, simg, vgg_str=str(tmp_path / "vgg.pth"), decoder_str=str(tmp_path / "dec.pth"),
                        content_size=64, style_size=48, alpha=0.5, output=str(tmp_path / "out"), file_name="x", save_ext=".png")
    assert p == tmp_path / "out" / "x.png" and p.exists()
    got = np.asarray(Image.open(p))
    assert got.shape == (64, 80, 3)
    # oracle on the same preprocessed tensors
    vgg_sd, dec_sd = weights
    ct = t.test_transform(64, False)(cimg).unsqueeze(0)
    stt = t.test_transform(48, False)(simg).unsqueeze(0)
    with torch.no_grad():
        ref = O.quantize_u8(O.style_transfer_simple(vgg_sd, dec_sd, ct, stt, 0.5))[0].numpy()
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and (got != ref).mean() < 0.01
    # mask + depth variant through the wrapper
    mask = (np.asarray(cimg.resize((80, 64))).transpose(2, 0, 1) > 40)
    p2 = adain_inference(cimg, simg, vgg_str=str(tmp_path / "vgg.pth"), decoder_str=str(tmp_path / "dec.pth"),
                         content_size=64, style_size=48, output=str(tmp_path / "out"), file_name="y", save_ext=".png",
                         content_mask=mask, use_depth=True, depth_map=T(synth.smooth_depth(63, 96, 120)), depth_offset=0.15)
    got2 = np.asarray(Image.open(p2))
    with torch.no_grad():
        o = O.style_transfer(vgg_sd, dec_sd, ct, stt, T(synth.smooth_depth(63, 96, 120)), 0.5, 0.15, 20)
        ref2 = O.quantize_u8(O.mask_composite(ct, o, T(mask)))[0].numpy()
    assert np.abs(got2.astype(int) - ref2.astype(int)).max() <= 1
    emb = get_style_embeddings(simg, vgg_str=str(tmp_path / "vgg.pth"), style_size=48)
    assert tuple(emb.shape) == (1, 512, 6, 8) and emb.is_cuda
    with torch.no_grad():
        close(emb, O.encode(vgg_sd, stt))


def test_edge_sizes_layouts_and_dtypes(rt, nets, weights):
    vgg, dec = nets
    vgg_sd, dec_sd = weights
    from applied_image_processing_amd.AdaIN import test as t

    # smallest legal image: 9x9 -> relu4_1 2x2 -> 16x16 (anything smaller cannot be reflection-padded at conv4_1)
    c, s = T(synth.image(91, 1, 9, 9)), T(synth.image(92, 1, 9, 12))
    out = t.style_transfer_simple(vgg, dec, c.cuda(), s.cuda(), 0.5)
    assert tuple(out.shape) == (1, 3, 16, 16)
    with torch.no_grad():
        close(out, O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5), 5e-4, 5e-4)
    with pytest.raises(rt.AdainHipError):
        vgg(torch.zeros(1, 3, 8, 64, device="cuda"))
    with pytest.raises(ValueError):
        vgg(torch.zeros(1, 4, 32, 32, device="cuda"))            # RGBA content is an error in the reference too (conv0 is 3->3)
    # non-contiguous and float64 inputs are accepted
    big = T(synth.image(93, 1, 40, 80)).cuda()
    view = big[:, :, :, ::2]
    assert not view.is_contiguous()
    f1 = vgg(view)
    f2 = vgg(view.contiguous().double())
    assert torch.equal(f1, f2)
    # the decoder takes either memory format and gives identical results
    assert f1.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(dec(f1), dec(f1.contiguous()))
    # width / height that are not multiples of the 32-pixel MFMA tile or the 8-row block
    c = T(synth.image(94, 2, 37, 99))
    with torch.no_grad():
        close(vgg(c.cuda()), O.encode(vgg_sd, c))


@pytest.mark.parametrize("n,hc,wc", [(1, 2, 2), (1, 2, 5), (2, 3, 9), (1, 5, 4), (1, 9, 3), (1, 4, 8), (1, 33, 17)])
def test_decoder_alone_vs_oracle(rt, weights, n, hc, wc):
    """The decoder on its own (net.py:6-36): feature maps whose 8x outputs end inside, on and past the last layer's 16 x 32
    tiles (one partial tile, several tiles across and down, a batch)."""
    _, dec_sd = weights
    feat = T(np.maximum(synth.uniform_sym(600 + hc * 40 + wc, (n, 512, hc, wc), 3.0), 0.0).astype(np.float32))   # relu4_1-like
    with torch.no_grad():
        ref = O.decode(dec_sd, feat)
    out = rt.decode(rt.nchw_to_nhwc(feat.cuda()), rt.pack_decoder(dec_sd, torch.device("cuda", 0)))
    assert tuple(out.shape) == (n, 3, 8 * hc, 8 * wc)
    close(out, ref, 5e-4, 5e-4)


def test_random_shapes_vs_oracle(rt, nets, weights):
    """Shape fuzz (a fixed seed; tests/fuzz_shapes.py runs more): random content / style sizes and batches - ragged tiles in
    both directions, several tiles across, tiny maps - through the alpha path, the depth path and the decoder alone."""
    from applied_image_processing_amd.AdaIN import test as t

    vgg, dec = nets
    vgg_sd, dec_sd = weights
    rng = np.random.default_rng(5)
    packed_dec = rt.pack_decoder(dec_sd, torch.device("cuda", 0))
    for case in range(24):
        n = int(rng.choice([1, 1, 2, 3]))
        h, w = int(rng.integers(9, 200)), int(rng.integers(9, 200))          # below 9 the reference fails too (pad of a 1-row map)
        if case % 8 == 7:
            w = int(rng.integers(500, 800))
        hs, ws = int(rng.integers(9, 120)), int(rng.integers(9, 120))
        c, s = T(synth.image(1000 + case, n, h, w)), T(synth.image(5000 + case, n, hs, ws))
        with torch.no_grad():
            if case % 3 == 0:
                alpha = float(rng.random())
                got, ref = t.style_transfer_simple(vgg, dec, c.cuda(), s.cuda(), alpha), O.style_transfer_simple(vgg_sd, dec_sd, c, s, alpha)
            elif case % 3 == 1 and n == 1:
                d = T(synth.smooth_depth(9000 + case, 2 * h + 3, w + 5))
                got = t.style_transfer(vgg, dec, c.cuda(), s.cuda(), d.cuda(), 1.0, 0.2, 15)
                ref = O.style_transfer(vgg_sd, dec_sd, c, s, d, 1.0, 0.2, 15)
            else:
                f = T(np.maximum(synth.uniform_sym(7000 + case, (n, 512, max(2, h // 8), max(2, w // 8)), 3.0), 0).astype(np.float32))
                got, ref = rt.decode(rt.nchw_to_nhwc(f.cuda()), packed_dec), O.decode(dec_sd, f)
        assert tuple(got.shape) == tuple(ref.shape)
        rel = float((got.cpu() - ref).norm() / ref.norm())
        assert rel <= 1e-4, (case, n, h, w, hs, ws, rel)


def test_called_from_a_background_thread(rt, nets):
    """The reference's callers run adain_inference on a background threading.Thread (GUI.py:131): a call from a thread
    that never touched the GPU gives the main thread's result bit for bit, and an error raised there carries that thread's text."""
    import threading
    from applied_image_processing_amd.AdaIN import test as t

    vgg, dec = nets
    c, s = T(synth.image(301, 1, 40, 72)).cuda(), T(synth.image(302, 1, 64, 48)).cuda()
    want = t.style_transfer_simple(vgg, dec, c, s, 0.6).cpu()
    got, errs = {}, {}

    def work(key):
        try:
            got[key] = t.style_transfer_simple(vgg, dec, c, s, 0.6).cpu()
            try:
                rt.conv3x3_wino(torch.zeros(1, 8, 8, 60, device="cuda"), torch.zeros(8, device="cuda"), torch.zeros(64, device="cuda"), 64, m_tiles=5)
            except rt.AdainHipError as e:
                errs[key] = str(e)
        except Exception as e:      # surfaced by the asserts below
            errs[key] = f"unexpected: {e!r}"

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
        th.join()
    for k in range(2):
        assert torch.equal(got[k], want)
        assert k in errs and not errs[k].startswith("unexpected") and "cin" in errs[k]


def test_batched_depth_and_mask_broadcast_rules(rt, weights):
    """Batch extensions of the C ABI (style_n / pmap_n / mask_n in {1, n}) against per-frame reference calls."""
    from applied_image_processing_amd.engine import AdaINEngine

    vgg_sd, dec_sd = weights
    e = AdaINEngine(vgg_sd, dec_sd, "cuda:0").set_style(T(synth.image(4, 1, 64, 64)).cuda())
    frames = T(np.concatenate([synth.image(120 + i, 1, 48, 80) for i in range(2)]))
    depths = [T(synth.smooth_depth(130 + i, 48, 80)) for i in range(2)]
    out = e.stylize_depth(frames.cuda(), [d.cuda() for d in depths], 0.2, 12)
    s = T(synth.image(4, 1, 64, 64))
    for i in range(2):
        with torch.no_grad():
            ref = O.style_transfer(vgg_sd, dec_sd, frames[i:i + 1], s, depths[i], 1.0, 0.2, 12)
        close(out[i:i + 1], ref, 5e-4, 5e-4)
    # per-frame masks [n,1,h,w] and one shared mask [1,3,h,w]
    m_n = (T(synth.image(140, 2, 20, 30, c=1)) > 0.5).float()
    m_1 = (T(synth.image(141, 1, 48, 80)) > 0.5).float()
    for m in (m_n, m_1):
        got = e.composite(frames.cuda(), out, m.cuda())
        for i in range(2):
            want = O.mask_composite(frames[i:i + 1], out[i:i + 1].cpu(), m[min(i, m.shape[0] - 1)])
            close(got[i:i + 1], want, 1e-5, 1e-5)
    with pytest.raises(rt.AdainHipError):
        rt.mask_composite(frames.cuda(), out, torch.zeros(2, 2, 48, 80, device="cuda"))   # mask channels must be 1 or 3


def test_statistics_edge_cases(rt):
    from applied_image_processing_amd.AdaIN import function as fn

    # one pixel per channel: torch's unbiased variance is NaN (0/0); the kernels reproduce that
    x = T(synth.uniform_sym(150, (1, 8, 1, 1), 1.0))
    m, s = fn.calc_mean_std(x.cuda())
    rm, rs = O.calc_mean_std(x)
    close(m, rm, 1e-6, 1e-7)
    assert torch.isnan(s).all() and torch.isnan(rs).all()
    # a large map through the multi-block NHWC reduction, both layouts bitwise identical run to run
    y = T(synth.uniform_sym(151, (1, 512, 96, 160), 3.0)).cuda().contiguous(memory_format=torch.channels_last)
    m1, s1 = fn.calc_mean_std(y)
    m2, s2 = fn.calc_mean_std(y)
    assert torch.equal(m1, m2) and torch.equal(s1, s2)
    rm, rs = O.calc_mean_std(y.cpu().contiguous())
    close(m1, rm, 1e-5, 1e-6)
    close(s1, rs, 1e-5, 1e-6)
    # strength map: large target, down- and up-scaling, against torch
    d = T(synth.smooth_depth(152, 300, 500))
    for size in ((600, 1000), (37, 61)):
        p = rt.strength_map(d.cuda(), size[0], size[1], 0.1, 15.0)
        close(p, O.compute_stylization_strength_map(d, size, 0.1, 15.0), 1e-4, 3e-5)


@pytest.mark.parametrize("sizes", [((1, 1000, 1016), (1, 500, 333)), ((2, 520, 392), (1, 264, 200), (1, 90, 70)), ((1, 64, 64), (1, 48, 80))])
def test_encode_multi_is_bitwise_the_separate_encodes(rt, weights, sizes):
    """adain_encode_multi: content batch and style image through the encoder in one pass, every 3x3 layer one launch whose tile
    list covers all segments (persistent kernel) - or one launch per segment when the work is too small.  Same tiles, same
    arithmetic: the features must equal the separate adain_encode results bit for bit (which are checked against the oracle)."""
    vgg_sd, _ = weights
    packed = rt.pack_encoder(vgg_sd, torch.device("cuda", 0))
    xs = [T(synth.image(800 + i, n, h, w)).cuda() for i, (n, h, w) in enumerate(sizes)]
    multi = rt.encode_multi(xs, packed)
    for x, f in zip(xs, multi):
        assert torch.equal(f, rt.encode(x, packed))
    with torch.no_grad():
        close(multi[-1].permute(0, 3, 1, 2), O.encode(vgg_sd, xs[-1].cpu()))
    with pytest.raises(rt.AdainHipError):
        rt.encode_multi(xs + xs + xs, packed)                     # more than four segments


def _conv_rows_cpu(x_nhwc_gpu, w, b, r0, r1, relu=True):
    """F.conv2d of reflect-padded rows r0..r1-1 of a (huge) NHWC GPU tensor, from the rows it needs only."""
    H = x_nhwc_gpu.shape[1]
    lo, hi = max(r0 - 1, 0), min(r1 + 1, H)
    crop = x_nhwc_gpu[:, lo:hi].cpu().permute(0, 3, 1, 2).contiguous()
    crop = F.pad(crop, (1, 1, 1 if r0 == 0 else 0, 1 if r1 == H else 0), mode="reflect")
    y = F.conv2d(crop, w, b)
    return (F.relu(y) if relu else y).permute(0, 2, 3, 1)


def test_conv_tensors_above_two_gib(rt):
    """Per-image tensors of 2 GiB and more (DCI-4K frames: 4096 x 2200 x 64 channels = 2.31 GB) take the per-tile buffer
    descriptors: rows at the top, around the 2 GiB mark (row 2048), and at the bottom against the CPU convolution of the rows
    they need; direct source, fused 2x upsample (small source, huge output) and fused output pool (huge source)."""
    H, W, C = 2200, 4096, 64
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.rand((1, H, W, C), device="cuda", generator=g) - 0.3
    assert x.numel() * 4 > 2 ** 31
    w = T(synth.uniform_sym(78, (C, C, 3, 3), (6.0 / (9 * C)) ** 0.5))
    b = T(synth.uniform_sym(79, (C,), 0.1))
    wp = rt.conv3x3_wino_pack(w.cuda(), 5)
    bands = ((0, 12), (2040, 2060), (H - 12, H))
    out = rt.conv3x3_wino(x, wp, b.cuda(), C, rt.SRC_DIRECT, True, False, 5)
    for r0, r1 in bands:
        close(out[:, r0:r1], _conv_rows_cpu(x, w, b, r0, r1))
    pooled = rt.conv3x3_wino(x, wp, b.cuda(), C, rt.SRC_DIRECT, True, True, 5)           # 1100 x 2048 x 64 out of the 2.31 GB source
    for r0, r1 in bands:
        ref = _conv_rows_cpu(x, w, b, r0, r1).permute(0, 3, 1, 2)
        close(pooled[:, r0 // 2:r1 // 2], F.max_pool2d(ref, 2, 2).permute(0, 2, 3, 1))
    del out, pooled
    small = x[:, :H // 2, :W // 2].contiguous()                                            # 1100 x 2048 -> 2x upsampled inside the conv
    up = rt.conv3x3_wino(small, wp, b.cuda(), C, rt.SRC_UP2X, True, False, 5)
    assert tuple(up.shape) == (1, H, W, C)
    for r0, r1 in bands:
        lo_s, hi_s = max(r0 - 1, 0) // 2, (min(r1 + 1, H) + 1) // 2                       # source rows under upsampled rows r0-1 .. r1
        u = small[:, lo_s:hi_s].repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)    # upsampled rows 2 lo_s .. 2 hi_s - 1
        close(up[:, r0:r1], _conv_rows_cpu(u, w, b, r0 - 2 * lo_s, r1 - 2 * lo_s))


def test_dci_4k_frame_bands_match_their_crops(rt, weights):
    """A 4096 x 2208 frame (9 Mpixels: the 64-channel layers hold 2.32 GB) through encoder and decoder: feature / image rows
    of the whole frame equal those of a 256-row band cut out of it, away from the cut (the bands start on multiples of 32 rows:
    pooling grids and Winograd tiles line up at every scale, so only the tile walk differs) - around the 2 GiB mark (row 2048)
    and at the bottom; a corner of the frame against the CPU oracle."""
    vgg_sd, dec_sd = weights
    enc, dec = rt.pack_encoder(vgg_sd, torch.device("cuda", 0)), rt.pack_decoder(dec_sd, torch.device("cuda", 0))
    H, W = 2208, 4096
    g = torch.Generator(device="cuda").manual_seed(91)
    img = torch.rand((1, 3, H, W), device="cuda", generator=g)
    full = rt.encode(img, enc)                                     # NHWC [1, 276, 512, 512]
    assert tuple(full.shape) == (1, 276, 512, 512)
    for a in (1920, H - 256):                                      # bands [a, a + 256): rows 2048.. cross 2 GiB in conv1_1 / conv1_2
        assert a % 32 == 0
        band = rt.encode(img[:, :, a:a + 256].contiguous(), enc)   # 32 feature rows
        m = 12 if a + 256 < H else 0                               # feature rows clear of the cut (receptive field < 96 pixels)
        close(full[:, a // 8 + 12:a // 8 + 32 - m], band[:, 12:32 - m], 1e-5, 1e-5)
    feat = torch.rand((1, 276, 512, 512), device="cuda", generator=g) * 2.0
    out = rt.decode(feat, dec)                                     # [1, 3, 2208, 4096]; dec8's output and conv_last's input: 2.32 GB
    assert tuple(out.shape) == (1, 3, H, W)
    for fa in (240, 276 - 32):
        assert fa % 4 == 0
        band = rt.decode(feat[:, fa:fa + 32].contiguous(), dec)    # 256 image rows
        m = 64 if fa + 32 < 276 else 0
        close(out[:, :, 8 * fa + 64:8 * (fa + 32) - m], band[:, :, 64:256 - m], 1e-5, 1e-5)
    # a corner of the whole frame against the CPU oracle (encoder: rows 0..255, far from the bottom)
    with torch.no_grad():
        ref = O.encode(vgg_sd, img[:, :, :384, :256].cpu())
    close(full[:, :24, :16].permute(0, 3, 1, 2), ref[:, :, :24, :16])
