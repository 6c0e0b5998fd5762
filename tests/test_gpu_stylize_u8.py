"""GPU tests of ``adain_stylize_u8`` (round 4): one sub-batch of decoded frames through the whole path in ONE call of the C ABI
must give the bytes of the separate calls it replaces (encode_u8 -> mean_std -> blend -> decode -> [composite] -> quantize_u8),
for the alpha and the depth-aware blend, with and without masks, in the fused tail (mask, frame and decoder output of one size)
and the general one (resizes), and against the CPU oracle.  Run with ``-m gpu``."""
import numpy as np
import pytest
import torch

import applied_image_processing_amd.synth as synth
from oracle import adain_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def u8frames(seed, n, h, w):
    return T(np.stack([(synth.image(seed + i, 1, h, w)[0].transpose(1, 2, 0) * 255).astype(np.uint8) for i in range(n)]))


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import applied_image_processing_amd.runtime as rt

    rt.lib()
    return rt


@pytest.fixture(scope="module")
def engine(weights):
    from applied_image_processing_amd.engine import AdaINEngine

    eng = AdaINEngine(weights[0], weights[1], "cuda:0")
    eng.set_style(T(synth.image(4, 1, 96, 128)).cuda())
    return eng


def separate_calls(engine, frames, alpha=0.5, depth=None, offset=0.15, prominence=20, masks=None):
    """The call sequence the job drivers ran before the single entry point existed."""
    if depth is not None:
        out = engine.stylize_depth(frames, depth, offset, prominence)
    else:
        out = engine.stylize(frames, alpha)
    if masks is not None:
        out = engine.composite(frames, out, masks.float())
    return engine.to_u8(out)


@pytest.mark.parametrize("n,h,w,alpha", [(1, 64, 96, 0.5), (2, 45, 67, 0.3), (3, 128, 72, 1.0), (1, 264, 200, 0.0)])
def test_alpha_blend_one_call_equals_separate_calls(engine, n, h, w, alpha):
    x = u8frames(700 + h, n, h, w).cuda()
    got = engine.stylize_u8(x, alpha=alpha)
    hc, wc = -(-h // 8), -(-w // 8)
    assert got.shape == (n, 8 * hc, 8 * wc, 3) and got.dtype == torch.uint8
    assert torch.equal(got, separate_calls(engine, x, alpha))
    buf = torch.zeros_like(got)
    assert engine.stylize_u8(x, alpha=alpha, out=buf) is buf and torch.equal(buf, got)


def test_depth_aware_one_call_equals_separate_calls_and_oracle(engine, weights):
    n, h, w = 2, 96, 136
    x = u8frames(720, n, h, w).cuda()
    depth = [T(synth.smooth_depth(6 + i, h0, w0)).cuda() for i, (h0, w0) in enumerate([(96, 136), (50, 70)])]     # maps of two sizes
    got = engine.stylize_u8(x, depth_maps=depth, offset=0.3, prominence=20)
    assert torch.equal(got, separate_calls(engine, x, depth=depth, offset=0.3))
    style = T(synth.image(4, 1, 96, 128))
    for i in range(n):
        c = x[i:i + 1].cpu().permute(0, 3, 1, 2).float().div(255)
        with torch.no_grad():
            ref = O.quantize_u8(O.style_transfer(weights[0], weights[1], c, style, depth[i].cpu(), 1.0, 0.3, 20))
        assert (got[i].cpu().int() - ref[0].int()).abs().max() <= 1


@pytest.mark.parametrize("mask_kind", ["bool3", "u8_1", "float3", "one_for_all"])
def test_masked_frames_fused_tail_equals_separate_calls(engine, mask_kind):
    n, h, w = 2, 64, 104                      # multiples of 8: mask, frame and decoder output share one size -> the fused kernel
    x = u8frames(740, n, h, w).cuda()
    if mask_kind == "bool3":
        m = (x > 60).permute(0, 3, 1, 2).contiguous()                       # train.py:97: view > 0 style mask, [n,3,h,w] bool
    elif mask_kind == "u8_1":
        m = (x[..., :1] > 90).permute(0, 3, 1, 2).to(torch.uint8).contiguous()      # localized_style_transfer.py:186: [n,1,h,w] uint8
    elif mask_kind == "float3":
        m = T(synth.image(9, n, h, w)).cuda()                               # fractional float mask: every rounding step matters
    else:
        m = (x[:1] > 128).permute(0, 3, 1, 2).contiguous()                  # one mask for the whole sub-batch
    got = engine.stylize_u8(x, alpha=0.5, masks=m)
    assert got.shape == (n, h, w, 3)
    assert torch.equal(got, separate_calls(engine, x, 0.5, masks=m))


@pytest.mark.parametrize("h,w,mh,mw", [(45, 67, 45, 67), (64, 104, 32, 52), (50, 70, 25, 35)])
def test_masked_frames_general_tail_equals_separate_calls(engine, h, w, mh, mw):
    """Frame sides that are not multiples of 8 (the decoder's output is larger: bilinear resize) and masks of another size
    (nearest resize): the tail of test.py:222-236 kernel by kernel."""
    n = 2
    x = u8frames(760 + h, n, h, w).cuda()
    m = (T(synth.image(11, n, mh, mw)) > 0.4).cuda()
    got = engine.stylize_u8(x, alpha=0.5, masks=m)
    assert got.shape == (n, h, w, 3)
    assert torch.equal(got, separate_calls(engine, x, 0.5, masks=m))
    mu8 = m.to(torch.uint8)
    assert torch.equal(engine.stylize_u8(x, alpha=0.5, masks=mu8), got)


def test_full_size_view_with_mask_against_oracle(engine, weights):
    h, w = 1200, 1600                         # BASELINE configs[4] view
    x = u8frames(780, 1, h, w)
    x[0][T(synth.uniform01(2000, h * w).reshape(h, w) < 0.3)] = 0
    xd = x.cuda()
    m = (xd > 0).permute(0, 3, 1, 2).contiguous()
    got = engine.stylize_u8(xd, alpha=0.5, masks=m)
    assert torch.equal(got, separate_calls(engine, xd, 0.5, masks=m))
    c = x.permute(0, 3, 1, 2).float().div(255)
    with torch.no_grad():
        ref = O.quantize_u8(O.mask_composite(c, O.style_transfer_simple(weights[0], weights[1], c, T(synth.image(4, 1, 96, 128)), 0.5), (c[0] > 0)))
    d = (got[0].cpu().int() - ref[0].int()).abs()
    assert d.max() <= 1 and float((d > 0).float().mean()) < 1e-3


def test_random_shapes_depth_and_mask_together(engine):
    """Seeded random sub-batches: frame sizes on and off the 8-pixel grid (both Winograd tile geometries, fused and general tails),
    1-3 frames, depth maps of their own sizes TOGETHER with masks of every dtype / channel count / batch form."""
    g = np.random.default_rng(20260404)
    for case in range(14):
        n = int(g.integers(1, 4))
        h, w = (int(g.integers(2, 20)) * 8, int(g.integers(2, 26)) * 8) if case % 2 == 0 else (int(g.integers(17, 150)), int(g.integers(17, 200)))
        x = u8frames(1000 + 10 * case, n, h, w).cuda()
        depth = None
        if case % 3 != 2:
            depth = [T(synth.smooth_depth(40 + case + i, int(g.integers(9, 90)), int(g.integers(9, 90)))).cuda() for i in range(n)]
        mh, mw = (h, w) if case % 4 < 2 else (int(g.integers(5, 60)), int(g.integers(5, 60)))
        mn, mc = (1 if case % 5 == 0 else n), (1 if case % 2 else 3)
        m = T(g.random((mn, mc, mh, mw)) > 0.45)
        m = [m, m.to(torch.uint8), m.float() * T(g.random((mn, mc, mh, mw)).astype(np.float32))][case % 3].cuda()
        got = engine.stylize_u8(x, alpha=0.4, depth_maps=depth, offset=0.25, prominence=12, masks=m)
        want = separate_calls(engine, x, 0.4, depth=depth, offset=0.25, prominence=12, masks=m)
        assert got.shape == (n, h, w, 3) and torch.equal(got, want), (case, n, h, w, mn, mc, mh, mw, str(m.dtype))


@pytest.mark.parametrize("n,h,w", [(3, 1080, 1920), (2, 1200, 1600), (2, 264, 1720), (4, 520, 776), (2, 1408, 1408),
                                   (2, 1440, 2560), (2, 2160, 3840)])
def test_batches_of_large_frames_two_phase_schedule_equals_frame_by_frame(rt, engine, n, h, w):
    """Batches of WIDE frames (>= 1600 pixels) run their big layers frame by frame and only the layers behind them over the whole batch
    (csrc/api.hip, BIG_FRAME_WIDTH / BIG_LAYER_ROUNDS; 1080p: everything but conv4_1 / dec1; a 264 x 1720 strip: the outer layers only),
    narrower ones every layer over the batch: either way the result must be, bit for bit, what each frame gives on its own - through
    the one-call entry point and through encode / decode on float frames.  1440 x 2560: conv4_1 alone is 7.2 rounds (it would count as
    big, but it writes the caller's feature tensor and therefore always runs over the batch: round-4 advisor finding, the feature
    tensor was never written); 2160 x 3840: every decoder layer is big (nothing runs over the batch)."""
    x = u8frames(1100 + h, n, h, w).cuda()
    m = (x > 40).permute(0, 3, 1, 2).contiguous()
    got = engine.stylize_u8(x, alpha=0.5, masks=m)
    for i in range(n):
        assert torch.equal(got[i:i + 1], engine.stylize_u8(x[i:i + 1].contiguous(), alpha=0.5, masks=m[i:i + 1].contiguous())), (n, h, w, i)
    f = rt.u8_to_f32(x)
    feats = rt.encode(f, engine.enc)
    for i in range(n):
        assert torch.equal(feats[i:i + 1], rt.encode(f[i:i + 1].contiguous(), engine.enc))
    out = rt.decode(feats, engine.dec)
    for i in range(n):
        assert torch.equal(out[i:i + 1], rt.decode(feats[i:i + 1].contiguous(), engine.dec))


def test_bad_arguments_raise(rt, engine):
    x = u8frames(790, 1, 32, 32).cuda()
    with pytest.raises(rt.AdainHipError):
        rt.stylize_u8(x.cpu(), engine.enc, engine.dec, engine.s_mean, engine.s_std)      # no CPU fallback
    with pytest.raises(rt.AdainHipError):
        rt.stylize_u8(x, engine.enc, engine.dec, engine.s_mean, engine.s_std, depth_maps=[])      # one map per frame
    with pytest.raises(rt.AdainHipError):
        engine.stylize_u8(x, masks=torch.zeros(1, 2, 32, 32, device="cuda"))  # mask channels 1 or 3
    with pytest.raises(rt.AdainHipError):
        engine.stylize_u8(u8frames(791, 1, 8, 8).cuda())                      # relu4_1 would be 1 x 1 (torch raises there too)
    with pytest.raises(AssertionError):
        engine.stylize_u8(x, alpha=1.5)                                      # test.py:75


def test_job_driver_uses_the_single_call_and_gives_the_same_frames(rt, engine):
    import applied_image_processing_amd.jobs as jobs

    n, h, w = 5, 64, 96
    frames = [f.numpy() for f in u8frames(800, n, h, w)]
    masks = [(f > 50).transpose(2, 0, 1) for f in frames]
    style = T(synth.image(4, 1, 96, 128)).cuda()
    before = rt.ABI_CALLS[0]
    out, info = jobs.stylize_frames_sharded(engine, frames, style, alpha=0.5, masks=masks, sub_batch=2, style_cache={0: engine.style_stats()})
    assert info["abi_calls"] == rt.ABI_CALLS[0] - before == 3             # three sub-batches, ONE C-ABI call each
    engine.use_style_stats(engine.style_stats())
    want = torch.cat([separate_calls(engine, T(np.stack(frames[i:i + 2])).cuda(), 0.5, masks=T(np.stack(masks[i:i + 2])).cuda())
                      for i in range(0, n, 2)])
    assert torch.equal(out, want)
    assert info["host_cpu_s"] >= 0 and info["process_cpu_s"] >= info["host_cpu_s"] * 0.5
