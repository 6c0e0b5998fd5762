"""GPU tests of the rows either side of the hot path: the INTER_AREA resize of the video post-pass, the
RCCL gather (world size 1: one process, one GPU), the sharded job drivers with the real engine, the depth-provider hook, the
CLI's pixels and the guide writer's file contents.  Run with ``-m gpu``."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import applied_image_processing_amd.synth as synth
from oracle import adain_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def u8img(seed, h, w, c=3):
    return (synth.image(seed, 1, h, w, c=c)[0].transpose(1, 2, 0) * 255).astype(np.uint8)


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import applied_image_processing_amd.runtime as rt

    rt.lib()
    return rt


@pytest.fixture(scope="module")
def engine(weights):
    from applied_image_processing_amd.engine import AdaINEngine

    vgg_sd, dec_sd = weights
    return AdaINEngine(vgg_sd, dec_sd, "cuda:0")


@pytest.fixture(scope="module")
def nccl_world1():
    """The default process group with RCCL as its CUDA backend, world size 1 (one process on one GPU is legal on the pool)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


# ---- INTER_AREA (video/utils.py:352-353) -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    (256, 456, 3, (256, 256)),     # the reference's own case: 456x256 stylised frame -> (256, 256): fractional x, y kept
    (256, 456, 3, (228, 128)),     # 2 x 2 fast form (vector kernel: width a multiple of 8)
    (1080, 1920, 3, (960, 540)),   # 1080p halved
    (100, 252, 3, (126, 50)),      # 2 x 2, width not a multiple of 8: the scalar form
    (6, 8, 3, (4, 3)),             # 2 x 2, one group of four output pixels per row
    (256, 456, 3, (152, 128)),     # integer 3 x 2 box
    (256, 456, 3, (456, 256)),     # same size: copy
    (256, 456, 3, (100, 77)),      # fractional both axes
    (255, 453, 3, (151, 85)),      # 3 x 3 box
    (90, 120, 1, (60, 45)),        # one channel, 2 x 2
    (90, 121, 4, (57, 31)),        # four channels, fractional
    (64, 64, 3, (1, 1)),           # everything into one pixel
    (1080, 1920, 3, (1280, 720)),  # 1080p -> 720p (scale 1.5)
    (256, 456, 3, (512, 288)),     # both axes enlarged (fractional): OpenCV's fixed-point linear emulation
    (64, 96, 3, (192, 128)),       # integer enlargement 2 x 2: replication
    (256, 456, 3, (256, 288)),     # x shrinks, y grows: still the linear path
    (90, 121, 4, (300, 31)),       # x grows, y shrinks, four channels
    (37, 53, 1, (54, 37)),         # one channel, x grows by 1 pixel
])
def test_resize_area_u8_bit_exact_vs_oracle(rt, case):
    hi, wi, c, dsize = case
    src = u8img(300 + hi + wi + c, hi, wi, c)
    got = rt.resize_area_u8(T(src).cuda(), dsize).cpu().numpy()
    want = O.resize_area_u8(src, dsize)
    if c == 1:
        want = want.reshape(got.shape)
    assert got.shape == (dsize[1], dsize[0], c)
    assert np.array_equal(got, want)


def test_resize_area_u8_random_sizes(rt):
    """Random source / destination sizes and channel counts: every branch (copy, 2 x 2, integer box, fractional taps, an
    enlarged axis, mixed) bit for bit against the oracle."""
    rng = np.random.default_rng(17)
    seen = set()
    for case in range(60):
        hi, wi, c = int(rng.integers(1, 90)), int(rng.integers(1, 130)), int(rng.choice([1, 3, 3, 4]))
        kind = case % 5
        if kind == 0:
            k = int(rng.integers(1, 4))
            ho, wo = max(1, hi // k), max(1, wi // k)
        elif kind == 1:
            ho, wo = int(rng.integers(1, hi + 1)), int(rng.integers(1, wi + 1))
        elif kind == 2:
            ho, wo = int(rng.integers(hi, 3 * hi + 2)), int(rng.integers(wi, 3 * wi + 2))
        elif kind == 3:
            ho, wo = int(rng.integers(1, hi + 1)), int(rng.integers(wi, 2 * wi + 2))
        else:
            ho, wo = hi * int(rng.integers(1, 4)), wi * int(rng.integers(1, 4))
        src = u8img(4000 + case, hi, wi, c)
        got = rt.resize_area_u8(T(src).cuda(), (wo, ho)).cpu().numpy()
        want = O.resize_area_u8(src, (wo, ho)).reshape(got.shape)
        assert np.array_equal(got, want), (hi, wi, c, ho, wo, int(np.abs(got.astype(int) - want.astype(int)).max()))
        seen.add((ho > hi or wo > wi, ho == hi and wo == wi))
    assert len(seen) >= 3


def test_resize_area_u8_batch_and_errors(rt):
    frames = np.stack([u8img(350 + i, 48, 80) for i in range(3)])
    got = rt.resize_area_u8(T(frames).cuda(), (30, 20)).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], O.resize_area_u8(frames[i], (30, 20)))
    half = rt.resize_area_u8(T(frames).cuda(), (40, 24)).cpu().numpy()           # 2 x 2 vector form, batched
    for i in range(3):
        assert np.array_equal(half[i], O.resize_area_u8(frames[i], (40, 24)))
    up = rt.resize_area_u8(T(frames).cuda(), (100, 48)).cpu().numpy()           # an enlarged axis, batched
    for i in range(3):
        assert np.array_equal(up[i], O.resize_area_u8(frames[i], (100, 48)))
    with pytest.raises(rt.AdainHipError):
        rt.resize_area_u8(T(frames).float().cuda(), (30, 20))


# ---- the one collective: gather over RCCL ----------------------------------------------------------------------------------------
def test_gather_frames_over_rccl_world1(rt, nccl_world1):
    import applied_image_processing_amd.sharding as sh

    frames = T(np.stack([u8img(360 + i, 40, 56) for i in range(5)])).cuda()
    assert sh.device_transport(frames) == "rccl"
    assert sh.device_transport(frames.cpu()) == "gloo"
    out = sh.gather_frames(frames, 5, dst=0)
    torch.cuda.synchronize()
    assert out.is_cuda and torch.equal(out, frames)
    fin = sh.gather_frames(frames, 5, dst=0, async_op=True)      # the overlapped form bench.py uses
    out2 = fin()
    torch.cuda.synchronize()
    assert torch.equal(out2, frames)
    with pytest.raises(ValueError):
        sh.gather_frames(frames[:4], 5, dst=0)


def test_video_job_real_engine_over_rccl(rt, engine, weights, nccl_world1):
    import applied_image_processing_amd.jobs as jobs

    vgg_sd, dec_sd = weights
    n, h, w = 4, 72, 128
    frames = [T(synth.image(370 + i, 1, h, w)[0]) for i in range(n)]
    styles = [T(synth.image(380, 1, 64, 64)), T(synth.image(381, 1, 48, 80))]
    depths = [T(synth.smooth_depth(385 + i, h, w)) for i in range(n)]
    flows = T(np.stack([synth.uniform_sym(390 + i, (2, 36, 64), 2.0) for i in range(n - 1)]))
    raw, info = jobs.stylize_frames_sharded(engine, frames, styles, style_of=jobs.style_schedule(n, 2), depth_maps=depths,
                                            depth_offset=0.30, depth_prominence=20, sub_batch=3, require_transport="rccl")
    assert info["transport"] is None and info["world"] == 1      # world 1: the block is the result, nothing to gather
    assert tuple(raw.shape) == (n, 72, 128, 3) and raw.dtype == torch.uint8
    sched = jobs.style_schedule(n, 2)
    assert sched == [0, 0, 1, 1]
    for i in range(n):                                            # stylisation: <= 1 LSB against the oracle, frame by frame
        with torch.no_grad():
            ref = O.quantize_u8(O.style_transfer(vgg_sd, dec_sd, frames[i][None], styles[sched[i]], depths[i], 1.0, 0.30, 20))[0]
        d = (raw[i].cpu().int() - ref.int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    out, info = jobs.video_style_transfer_sharded(engine, frames, styles, flows=flows.cuda(), target_resolution=(64, 36), blend_alpha=0.7,
                                                  depth_maps=depths, sub_batch=3)
    # the post-pass is integer / uncontracted fp32: bit-exact against the oracle fed with the GPU's own stylised frames
    small = np.stack([O.resize_area_u8(f, (64, 36)) for f in raw.cpu().numpy()])
    assert np.array_equal(out.cpu().numpy(), O.temporal_blend(small, flows.numpy(), 0.7))
    assert "temporal_blend_s" in info


def test_guides_job_real_engine_file_contents(rt, engine, weights, nccl_world1, tmp_path):
    from PIL import Image

    import applied_image_processing_amd.jobs as jobs
    from applied_image_processing_amd.AdaIN.test import test_transform
    from applied_image_processing_amd.engine import precompute_guides

    vgg_sd, dec_sd = weights
    style = T(synth.image(400, 1, 64, 64))
    pil = [Image.fromarray(u8img(410 + i, 60, 90)) for i in range(3)]
    names = ["r_0", "r_1", "r_2"]
    tf = test_transform(48, False)
    masks = [np.asarray(p.resize((72, 48))).transpose(2, 0, 1) > 60 for p in pil]
    for write in ("dst", "local"):
        paths, info = jobs.precompute_guides_sharded(engine, pil, names, tmp_path / write, style, masks=masks, content_size=48,
                                                     save_ext=".png", write=write, sub_batch=2)
        assert [paths[n].name for n in names] == ["r_0.png", "r_1.png", "r_2.png"]        # train.py:104-114 naming
        for k, nm in enumerate(names):
            c = tf(pil[k])[None]
            with torch.no_grad():
                ref = O.quantize_u8(O.mask_composite(c, O.style_transfer_simple(vgg_sd, dec_sd, c, style, 0.5), T(masks[k])))[0].numpy()
            got = np.asarray(Image.open(paths[nm]))
            assert got.shape == ref.shape == (48, 72, 3)
            assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and (got != ref).mean() < 0.01
    # the single-rank writer produces the same files
    engine.set_style(style.cuda())
    p1 = precompute_guides(engine, pil, names, tmp_path / "single", masks=masks, content_size=48, save_ext=".png", sub_batch=2)
    for nm in names:
        assert np.array_equal(np.asarray(Image.open(p1[nm])), np.asarray(Image.open(tmp_path / "dst" / f"{nm}.png")))


# ---- host hooks ---------------------------------------------------------------------------------------------------------------------
def test_depth_provider_hook_and_cli_pixels(rt, weights, tmp_path):
    from PIL import Image

    from applied_image_processing_amd.AdaIN import run_depth, test as t

    vgg_sd, dec_sd = weights
    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(dec_sd, tmp_path / "dec.pth")
    cimg, simg = Image.fromarray(u8img(420, 80, 96)), Image.fromarray(u8img(421, 64, 64))
    cimg.save(tmp_path / "c.png")
    simg.save(tmp_path / "s.png")
    depth = synth.smooth_depth(422, 80, 96)
    np.save(tmp_path / "d.npy", depth)
    ct, stt = t.test_transform(512, False)(cimg)[None], t.test_transform(512, False)(simg)[None]
    with torch.no_grad():
        ref = O.quantize_u8(O.style_transfer(vgg_sd, dec_sd, ct, stt, T(depth), 0.5, 0.15, 20))[0].numpy()
    # (1) the CLI with a precomputed map; PNG so that the file holds the quantised pixels themselves
    seen = []

    def provider(img):
        seen.append(img.size if hasattr(img, "size") else None)
        return T(depth)

    common = ["--content", str(tmp_path / "c.png"), "--style", str(tmp_path / "s.png"), "--output", str(tmp_path / "o"), "--use_depth",
              "--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "dec.pth")]
    p = run_depth.main(common + ["--depth_npy", str(tmp_path / "d.npy"), "--file_name", "a"])
    jpg = np.asarray(Image.open(p))
    assert p.name == "a.jpg" and jpg.shape == ref.shape == (512, 616, 3)
    assert float(np.abs(jpg.astype(float) - ref.astype(float)).mean()) < 6.0            # JPEG (quality 75) of the same pixels
    # (2) the provider hook: called once with the content PIL image, same pixels as the precomputed map
    t.set_depth_provider(provider)
    try:
        p2 = t.adain_inference(str(tmp_path / "c.png"), str(tmp_path / "s.png"), vgg_str=str(tmp_path / "vgg.pth"),
                               decoder_str=str(tmp_path / "dec.pth"), depth_offset=0.15, output=str(tmp_path / "o"), file_name="b",
                               save_ext=".png", use_depth=True)
        assert seen == [(96, 80)]
        got = np.asarray(Image.open(p2))
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and (got != ref).mean() < 0.01
    finally:
        t.set_depth_provider(None)
    assert t._depth_provider is None


# ---- Winograd F(4,3) x F(2,3) where its transforms cancel: large DC inputs, zero-sum filters ------------------------------------------
@pytest.mark.parametrize("cin,cout,hw", [(64, 64, (40, 72)), (256, 256, (24, 40)), (512, 256, (16, 24))])
def test_winograd_large_dc_zero_sum_filters(rt, cin, cout, hw):
    """Post-ReLU-like inputs (non-negative, mean 50, sigma 1) against filters whose nine taps sum to zero: the exact result
    is O(1) while the F(4,3) input transform (coefficients up to 5) works on values of magnitude 50 x 10.  The error bound
    asserted is the path's per-layer tolerance (2e-4 absolute + relative) scaled by nothing: it must hold as is."""
    import torch.nn.functional as F

    h, w = hw
    x = 50.0 + T(synth.uniform_sym(700 + cin, (1, cin, h, w), 3.0 ** 0.5))          # uniform with sigma 1
    assert float(x.min()) > 0
    wt = T(synth.uniform_sym(710 + cout, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5))
    wt = wt - wt.mean(dim=(2, 3), keepdim=True)                                     # every 3x3 filter sums to zero
    b = T(synth.uniform_sym(720 + cout, (cout,), 0.1))
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), wt.double(), b.double())
    xg = x.cuda().permute(0, 2, 3, 1).contiguous()
    errs = {}
    packed = rt.conv3x3_wino_pack(wt.cuda(), 5)
    out = rt.conv3x3_wino(xg, packed, b.cuda(), cout, rt.SRC_DIRECT, relu=False, m_tiles=5).permute(0, 3, 1, 2).cpu().double()
    errs[5] = float((out - ref).abs().max())
    cpu32 = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt, b).double()
    errs["torch_cpu_fp32"] = float((cpu32 - ref).abs().max())
    print(f"large-DC cin={cin}: max abs error vs fp64 (|ref| max {float(ref.abs().max()):.2f}): {errs}")
    assert errs[5] <= 2e-4 + 2e-4 * float(ref.abs().max())


def test_localized_pipeline_end_to_end(rt, weights, tmp_path):
    """run_localized_style_transfer (reference Style_3DGS/localized_style_transfer.py:191-245) through the HIP path: the mask
    provider hook replaces DeepLabV3, the background is stylised with alpha = 1 under the mask, the foreground keeps its
    pixels up to the colour transfer."""
    from PIL import Image

    from applied_image_processing_amd import localized as L

    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(weights[1], tmp_path / "dec.pth")
    Image.fromarray(u8img(430, 64, 96)).save(tmp_path / "c.png")
    Image.fromarray(u8img(431, 64, 64)).save(tmp_path / "s.png")
    yy, xx = np.mgrid[:64, :96]
    bgmask = (((yy - 30) ** 2 + (xx - 50) ** 2) > 300).astype(np.uint8)[None]          # [1,H,W], 1 = background
    calls = []
    L.set_mask_provider(lambda img: (calls.append(img.size), bgmask)[1])
    try:
        p = L.run_localized_style_transfer(str(tmp_path / "c.png"), str(tmp_path / "s.png"), output_path=str(tmp_path / "o"),
                                           file_name="loc", vgg_str=str(tmp_path / "vgg.pth"), decoder_str=str(tmp_path / "dec.pth"),
                                           content_size=0, save_ext=".png")
    finally:
        L.set_mask_provider(None)
    assert calls == [(96, 64)] and p == f"{tmp_path / 'o'}/localized_style_transfer_result.jpg"
    sty = np.asarray(Image.open(tmp_path / "o" / "loc.png"))
    content = np.asarray(Image.open(tmp_path / "c.png"))
    assert np.array_equal(sty[bgmask[0] == 0], content[bgmask[0] == 0])                # adain_inference kept the foreground
    want = L.combine_localized(content, sty, bgmask[0])
    got = np.asarray(Image.open(p))
    assert got.shape == want.shape and float(np.abs(got.astype(float) - want.astype(float)).mean()) < 8.0    # JPEG of `want`
    with pytest.raises(RuntimeError, match="provider"):
        L.run_localized_style_transfer(str(tmp_path / "c.png"), str(tmp_path / "s.png"), output_path=str(tmp_path / "o"))


def test_run_semantic_segm_cli_equals_the_function_call(rt, weights, tmp_path):
    """The CLI of the reference's Style_3DGS/run_semantic_segm.py (:12-44: --content --style --output --file_name --use_depth) with a
    precomputed mask and proximity map in place of the two network downloads: same files as calling the pipeline directly."""
    from PIL import Image

    from applied_image_processing_amd import localized as L
    from applied_image_processing_amd import run_semantic_segm as cli

    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(weights[1], tmp_path / "dec.pth")
    Image.fromarray(u8img(440, 72, 104)).save(tmp_path / "c.png")
    Image.fromarray(u8img(441, 64, 64)).save(tmp_path / "s.png")
    yy, xx = np.mgrid[:72, :104]
    bg = (((yy - 36) ** 2 + (xx - 52) ** 2) > 400).astype(np.uint8)
    np.save(tmp_path / "mask.npy", bg)                                              # [H,W]: the CLI adds the channel axis
    np.save(tmp_path / "depth.npy", synth.smooth_depth(7, 72, 104))
    ck = ["--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "dec.pth")]
    p = cli.main(["--content", str(tmp_path / "c.png"), "--style", str(tmp_path / "s.png"), "--output", str(tmp_path / "cli"), "--use_depth",
                  "--mask_npy", str(tmp_path / "mask.npy"), "--depth_npy", str(tmp_path / "depth.npy")] + ck)
    q = L.run_localized_style_transfer(str(tmp_path / "c.png"), str(tmp_path / "s.png"), output_path=str(tmp_path / "fn"), file_name="stylized",
                                       use_depth=True, background_mask=bg[None], depth_map=T(synth.smooth_depth(7, 72, 104)),
                                       vgg_str=str(tmp_path / "vgg.pth"), decoder_str=str(tmp_path / "dec.pth"))
    assert p == f"{tmp_path / 'cli'}/localized_style_transfer_result.jpg" and (tmp_path / "cli" / "stylized.jpg").exists()     # reference naming
    assert open(p, "rb").read() == open(q, "rb").read()
    assert (tmp_path / "cli" / "stylized.jpg").read_bytes() == (tmp_path / "fn" / "stylized.jpg").read_bytes()
    with pytest.raises(SystemExit):
        cli.main(["--style", "s.png"])                                              # --content is required, as in the reference


def test_checkpoint_cache_notices_foreign_weights(rt, weights, tmp_path):
    """adain_inference caches the loaded checkpoints per file version; weights written into the module singletons by anyone
    else in between must not be mistaken for the file's (results must not depend on call order)."""
    from PIL import Image

    from applied_image_processing_amd.AdaIN import net, test as t

    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(weights[1], tmp_path / "dec.pth")
    cimg, simg = Image.fromarray(u8img(440, 40, 48)), Image.fromarray(u8img(441, 32, 32))
    kw = dict(vgg_str=str(tmp_path / "vgg.pth"), decoder_str=str(tmp_path / "dec.pth"), content_size=0, style_size=0,
              output=str(tmp_path / "o"), save_ext=".png")
    a = np.asarray(Image.open(t.adain_inference(cimg, simg, file_name="a", **kw)))
    net.decoder.load_state_dict(synth.to_torch(synth.decoder_state_dict(7)))       # someone else's weights
    b = np.asarray(Image.open(t.adain_inference(cimg, simg, file_name="b", **kw)))
    assert np.array_equal(a, b)
    with torch.no_grad():
        net.vgg[2].weight.mul_(0.5)                                                  # in-place edit
    c = np.asarray(Image.open(t.adain_inference(cimg, simg, file_name="c", **kw)))
    assert np.array_equal(a, c)


def test_video_caller_from_directories(rt, engine, weights, tmp_path):
    """video.apply_style_transfer_multi_ada (reference video/utils.py:297-372) from frame / style directories with the depth and
    flow providers plugged: output files keep the frame names, frame 0 is the stylised + resized frame, later frames the
    recurrence; compared with the oracle chain (<= 2 levels: a 1-level difference of a stylised frame can move a blended pixel)."""
    from PIL import Image

    import applied_image_processing_amd.jobs as jobs
    from applied_image_processing_amd import video
    from applied_image_processing_amd.AdaIN import test as t

    vgg_sd, dec_sd = weights
    cdir, sdir, odir = tmp_path / "frames", tmp_path / "styles", tmp_path / "out"
    cdir.mkdir(); sdir.mkdir()
    n = 4
    for i in range(n):
        Image.fromarray(u8img(450 + i, 72, 128)).save(cdir / f"frame_{i:04d}.png")
    for i in range(2):
        Image.fromarray(u8img(460 + i, 64, 64)).save(sdir / f"style_{i}.png")
    flows = [synth.uniform_sym(470 + i, (2, 36, 64), 2.0) for i in range(n - 1)]
    calls = []

    def flow_provider(prev_path, cur_path, target_resolution, method):
        calls.append((os.path.basename(prev_path), os.path.basename(cur_path), tuple(target_resolution), method))
        return flows[len(calls) - 1]

    t.set_depth_provider(lambda img: T(synth.smooth_depth(480 + img.size[0] % 7, img.size[1], img.size[0])))
    video.set_flow_provider(flow_provider)
    try:
        out = video.apply_style_transfer_multi_ada(str(cdir), str(sdir), str(odir), flow_method="dualtvl1", alpha=0.7,
                                                   target_resolution=(64, 36), engine=engine)
    finally:
        t.set_depth_provider(None)
        video.set_flow_provider(None)
    assert out == odir and sorted(os.listdir(odir)) == [f"frame_{i:04d}.png" for i in range(n)]
    assert calls == [(f"frame_{i:04d}.png", f"frame_{i + 1:04d}.png", (64, 36), "dualtvl1") for i in range(n - 1)]
    # oracle chain
    tf, stf = t.test_transform(256, False), t.test_transform(512, False)
    styles = [stf(Image.open(sdir / f"style_{i}.png")).unsqueeze(0) for i in range(2)]
    sched = jobs.style_schedule(n, 2)
    depth = T(synth.smooth_depth(480 + 128 % 7, 72, 128))
    small = []
    for i in range(n):
        c = tf(Image.open(cdir / f"frame_{i:04d}.png")).unsqueeze(0)
        with torch.no_grad():
            u8 = O.quantize_u8(O.style_transfer(vgg_sd, dec_sd, c, styles[sched[i]], depth, 1.0, 0.30, 20))[0].numpy()
        small.append(O.resize_area_u8(u8, (64, 36)))
    want = O.temporal_blend(np.stack(small), np.stack(flows), 0.7)
    for i in range(n):
        got = np.asarray(Image.open(odir / f"frame_{i:04d}.png"))
        d = np.abs(got.astype(int) - want[i].astype(int))
        assert got.shape == (36, 64, 3) and d.max() <= 2 and (d > 0).mean() < 0.02, (i, d.max(), (d > 0).mean())
    with pytest.raises(RuntimeError, match="provider"):
        video.estimate_optical_flow("a", "b", (64, 36))
