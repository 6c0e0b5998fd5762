"""GPU tests of the job drivers' host side (round 3): ToTensor on the device (``adain_u8_to_f32``, ``adain_encode_u8``) bit for bit
against the host's ``/ 255``, the frame feeder (pinned staging + copy stream) against device-resident inputs, the asynchronous
file sink with views of mixed sizes.  Run with ``-m gpu``."""
import numpy as np
import pytest
import torch

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

    return AdaINEngine(weights[0], weights[1], "cuda:0")


def test_u8_to_f32_is_totensor_bit_for_bit(rt):
    """Every uint8 value and every launch form (RGB x 4-pixel vector form, generic channel counts, ragged sizes) against the
    host's ``tensor.float() / 255`` (torchvision ToTensor, reference test.py:22)."""
    ramp = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).expand(1, 16, 16, 3).contiguous()
    got = rt.u8_to_f32(ramp.cuda()).cpu()
    assert torch.equal(got, ramp.permute(0, 3, 1, 2).float().div(255))
    for (n, h, w, c) in [(2, 8, 12, 3), (1, 7, 9, 3), (3, 5, 6, 1), (1, 11, 13, 4), (2, 1080, 1920, 3)]:
        x = T(np.stack([u8img(500 + i + h, h, w, c) for i in range(n)]))
        got = rt.u8_to_f32(x.cuda()).cpu()
        assert got.shape == (n, c, h, w)
        assert torch.equal(got, x.permute(0, 3, 1, 2).float().div(255))
    with pytest.raises(rt.AdainHipError):
        rt.u8_to_f32(torch.zeros(1, 4, 4, 3).cuda())


@pytest.mark.parametrize("shape", [(1, 64, 96), (2, 45, 67), (1, 256, 256), (3, 33, 40)])
def test_encode_u8_equals_encode_of_totensor_bitwise(rt, engine, shape):
    n, h, w = shape
    x = T(np.stack([u8img(520 + i, h, w) for i in range(n)]))
    host_f32 = x.permute(0, 3, 1, 2).float().div(255).contiguous()
    want = rt.encode(host_f32.cuda(), engine.enc)
    assert torch.equal(rt.encode_u8(x.cuda(), engine.enc), want)
    assert torch.equal(rt.encode(rt.u8_to_f32(x.cuda()), engine.enc), want)
    with pytest.raises(rt.AdainHipError):
        rt.encode_u8(host_f32.cuda(), engine.enc)
    with pytest.raises(rt.AdainHipError):
        rt.encode_u8(x, engine.enc)                                # a host tensor: no CPU fallback


def test_job_from_host_frames_equals_device_resident_floats(rt, engine):
    """The same job three ways: float frames already on the GPU (round 2's form), decoded uint8 frames in pageable host memory
    through the feeder (pinned slots, copy stream, ToTensor on the device) with masks and proximity maps riding along, and the
    zero-copy ``block`` protocol of a device-resident uint8 store: identical uint8 results."""
    import applied_image_processing_amd.jobs as jobs

    n, h, w = 7, 72, 104
    u8 = [u8img(540 + i, h, w) for i in range(n)]
    f32_dev = [T(f).permute(2, 0, 1).float().div(255).cuda() for f in u8]
    masks = [(f > 70).transpose(2, 0, 1) for f in u8]
    depths = [synth.smooth_depth(560 + i, h, w) for i in range(n)]
    styles = [T(synth.image(570, 1, 64, 64)), T(synth.image(571, 1, 48, 80))]
    sched = jobs.style_schedule(n, 2)
    ref, _ = jobs.stylize_frames_sharded(engine, f32_dev, styles, style_of=sched, masks=[T(m).cuda() for m in masks], sub_batch=3)
    host = torch.zeros((n, h, w, 3), dtype=torch.uint8).pin_memory()
    got, info = jobs.stylize_frames_sharded(engine, u8, styles, style_of=sched, masks=masks, sub_batch=3, host_out=host)
    assert info["h2d_bytes"] >= n * h * w * 3 and torch.equal(got, ref)
    assert info["d2h_bytes"] == n * h * w * 3 and torch.equal(host, ref.cpu())       # the asynchronous copies have all landed
    refd, _ = jobs.stylize_frames_sharded(engine, f32_dev, styles, style_of=sched, depth_maps=[T(d).cuda() for d in depths], sub_batch=2)
    gotd, _ = jobs.stylize_frames_sharded(engine, u8, styles, style_of=sched, depth_maps=depths, sub_batch=2, prefetch=2)
    assert torch.equal(gotd, refd) and not torch.equal(gotd, got)

    class Store:                                                   # device-resident decoded frames: views, no staging
        dev = T(np.stack(u8)).cuda()

        def __len__(self):
            return n

        def __getitem__(self, k):
            return self.dev[k]

        def block(self, i, j):
            return self.dev[i:j]

    gotb, infob = jobs.stylize_frames_sharded(engine, Store(), styles, style_of=sched, masks=masks, sub_batch=3)
    assert torch.equal(gotb, ref)
    # against the oracle: <= 1 LSB
    vgg_sd, dec_sd = engine_weights()
    with torch.no_grad():
        c = f32_dev[4].cpu()[None]
        want = O.quantize_u8(O.mask_composite(c, O.style_transfer_simple(vgg_sd, dec_sd, c, styles[sched[4]], 0.5), T(masks[4])))[0]
    d = (got[4].cpu().int() - want.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01


def engine_weights():
    return synth.to_torch(synth.vgg_state_dict(0, full=False)), synth.to_torch(synth.decoder_state_dict(0))


def test_feeder_error_reaches_the_caller(rt, engine):
    import applied_image_processing_amd.jobs as jobs

    class Bad:
        def __len__(self):
            return 4

        def __getitem__(self, k):
            if k == 2:
                raise OSError("cannot decode frame 2")
            return u8img(580 + k, 40, 56)

    with pytest.raises(OSError, match="frame 2"):
        jobs.stylize_frames_sharded(engine, Bad(), T(synth.image(581, 1, 48, 48)), sub_batch=1)
    out, _ = jobs.stylize_frames_sharded(engine, [u8img(580, 40, 56)], T(synth.image(581, 1, 48, 48)))      # the engine is still usable
    assert tuple(out.shape) == (1, 40, 56, 3)


def test_guides_local_writer_mixed_view_sizes(rt, engine, tmp_path):
    """write="local": views of different sizes in one job (mixed cameras), files written by the asynchronous sink; every file
    <= 1 LSB from the oracle."""
    from PIL import Image

    import applied_image_processing_amd.jobs as jobs
    from applied_image_processing_amd.AdaIN.test import test_transform

    vgg_sd, dec_sd = engine_weights()
    style = T(synth.image(590, 1, 64, 64))
    sizes = [(60, 90), (60, 90), (48, 48), (80, 56), (60, 90)]
    pil = [Image.fromarray(u8img(600 + i, hh, ww)) for i, (hh, ww) in enumerate(sizes)]
    names = [f"cam_{i}" for i in range(len(pil))]
    paths, info = jobs.precompute_guides_sharded(engine, pil, names, tmp_path / "g", style, content_size=40, save_ext=".png",
                                                 write="local", sub_batch=2)
    assert info["d2h_bytes"] > 0
    tf = test_transform(40, False)
    for k, nm in enumerate(names):
        c = tf(pil[k])[None]
        with torch.no_grad():
            ref = O.quantize_u8(O.style_transfer_simple(vgg_sd, dec_sd, c, style, 0.5))[0].numpy()
        got = np.asarray(Image.open(paths[nm]))
        assert got.shape == ref.shape
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and (got != ref).mean() < 0.01
    with pytest.raises(ValueError):
        jobs.precompute_guides_sharded(engine, pil, names, tmp_path / "g2", style, content_size=40, save_ext=".png", write="dst")


def test_bench_frame_store_is_the_host_generator(rt):
    """bench.py --job builds its frames on the GPU; they are the host generator's bits (SURVEY.md 8(d) seeds)."""
    f = synth.frame_u8_torch(7 + 3, 36, 64, "cuda:0").cpu().numpy()
    img = synth.image(7 + 3, 1, 36, 64)[0]
    assert np.array_equal(f, np.floor(img * 256).astype(np.uint8).transpose(1, 2, 0))
    v = synth.frame_u8_torch(1000, 36, 64, "cuda:0", zero_fraction=0.3, zero_seed=2000).cpu().numpy()
    bg = synth.uniform01(2000, 36 * 64).reshape(36, 64) < 0.3
    assert (v[bg] == 0).all() and 0.2 < bg.mean() < 0.4


def test_bench_depth_store_is_the_host_generator_and_job_equals_direct_call(rt, engine):
    """bench.py --config 4 --depth: the per-frame proximity maps built on the GPU are ``synth.smooth_depth``'s bits, and the job
    driver's depth-aware frames equal direct depth-aware engine calls ("a per-frame synthetic depth")."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import bench
    import applied_image_processing_amd.jobs as jobs

    h, w, n = 64, 96, 5
    frames = bench.FrameStore(4, n, 0, n, h, w, torch.device("cuda", 0))
    depths = bench.DepthStore(frames, h, w, torch.device("cuda", 0))
    for k in (0, 3):
        assert np.array_equal(depths[k].cpu().numpy(), synth.smooth_depth(6 + k, h, w))
    style = torch.from_numpy(synth.image(4, 1, 64, 64)).cuda()
    u8, _ = jobs.stylize_frames_sharded(engine, frames, style, depth_maps=depths, depth_offset=0.30, depth_prominence=20, sub_batch=2, gather=False)
    engine.set_style(style)
    for k in range(n):
        direct = engine.to_u8(engine.stylize_depth(frames[k][None], [depths[k]], 0.30, 20))
        assert torch.equal(u8[k], direct[0]), k


def test_jobs_random_shapes_host_feeder_vs_direct_engine_calls(rt, engine):
    """Random frame counts, sizes (both tile geometries, ragged), sub-batch sizes, feeder depths, with / without masks and proximity
    maps: the job driver fed from pageable host uint8 frames must give exactly the uint8 frames of direct engine calls on
    device-resident float tensors, frame by frame."""
    import applied_image_processing_amd.jobs as jobs

    rng = np.random.default_rng(23)
    style = T(synth.image(700, 1, 64, 80)).cuda()
    stats = engine.set_style(style).style_stats()
    for case in range(14):
        n, h, w = int(rng.integers(1, 7)), int(rng.integers(9, 130)), int(rng.integers(9, 210))
        sub, depth_slots = (None if case % 4 == 3 else int(rng.integers(1, 5))), int(rng.integers(2, 5))      # None: the driver's automatic size
        kind = case % 3                                                     # 0 plain, 1 masks, 2 proximity maps
        u8 = [u8img(710 + 10 * case + i, h, w) for i in range(n)]
        masks = [(f > 90).transpose(2, 0, 1) for f in u8] if kind == 1 else None
        depths = [synth.smooth_depth(720 + 10 * case + i, h + 3, w + 5) for i in range(n)] if kind == 2 else None
        got, info = jobs.stylize_frames_sharded(engine, u8, style, masks=masks, depth_maps=depths, depth_offset=0.2, depth_prominence=15,
                                                sub_batch=sub, prefetch=depth_slots, style_cache={0: stats}, fetch_workers=int(rng.integers(1, 4)))
        engine.use_style_stats(stats)
        for i in range(n):
            c = T(u8[i]).permute(2, 0, 1).float().div(255).unsqueeze(0).cuda()
            if kind == 2:
                out = engine.stylize_depth(c, [T(depths[i]).cuda()], 0.2, 15)
            else:
                out = engine.stylize(c, 0.5)
            if kind == 1:
                out = engine.composite(c, out, T(masks[i]).float().unsqueeze(0).cuda())
            want = engine.to_u8(out)[0]
            assert torch.equal(got[i], want), (case, n, h, w, sub, kind, i)
        assert info["feeder"]["batches"] >= -(-n // (sub or 32))


def test_job_under_a_side_stream_with_frames_made_on_the_fetch_threads(rt, engine):
    """The guide / video feeders hand over frames that the FETCH threads made on the device (device_transform_u8: the resized frame is
    allocated under the fetching thread's stream).  Run under ``torch.cuda.stream(side)`` the consumer reads them on another stream:
    the feeder now tells the allocator so (``record_stream``); the job must give the bytes of the default-stream run, pass after pass,
    while later fetches keep allocating (round-5 advisor finding)."""
    from PIL import Image

    import applied_image_processing_amd.jobs as jobs
    from applied_image_processing_amd.AdaIN.test import device_transform_u8

    dev = torch.device("cuda:0")
    pil = [Image.fromarray(u8img(900 + i, 120 + 8 * (i % 3), 200)) for i in range(24)]

    class Views:                          # every fetch decodes nothing but resizes on the device: a fresh device tensor per frame
        def __len__(self):
            return len(pil)

        def __getitem__(self, k):
            return device_transform_u8(pil[k], 96, False, dev)[0]

    style = T(synth.image(700, 1, 64, 80)).cuda()
    stats = engine.set_style(style).style_stats()
    want, _ = jobs.stylize_frames_sharded(engine, Views(), style, sub_batch=3, style_cache={0: stats}, gather=False, fetch_workers=3)
    want = [w.clone() for w in want]
    side = torch.cuda.Stream(dev)
    for _ in range(3):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            got, _ = jobs.stylize_frames_sharded(engine, Views(), style, sub_batch=3, style_cache={0: stats}, gather=False, fetch_workers=3)
            side.synchronize()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
