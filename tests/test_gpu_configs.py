"""GPU tests at the full sizes of BASELINE.json configs 3-5 (config 2 is in test_gpu_parity.py): one unit per
config against the CPU oracle (relative L2 <= 1e-4, PSNR >= 60 dB after the [0,1] clamp) plus size-independent
properties for the batched paths (a batch equals its frames run one by one, bitwise; uint8 quantiser bit-exact)."""
import numpy as np
import pytest
import torch

import applied_image_processing_amd.synth as synth
from oracle import adain_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def engine(weights):
    from applied_image_processing_amd.engine import AdaINEngine

    vgg_sd, dec_sd = weights
    e = AdaINEngine(vgg_sd, dec_sd, "cuda:0")
    e.set_style(T(synth.image(4, 1, 512, 512)).cuda())
    return e


def check(out, ref, what):
    out = out.cpu()
    rel = float((out - ref).norm() / ref.norm())
    psnr = float(O.psnr(out.clamp(0, 1), ref.clamp(0, 1)).min())
    print(f"{what}: relative L2 {rel:.3e}, PSNR {psnr:.1f} dB")
    assert rel <= 1e-4 and psnr >= 60.0


# ---- BASELINE sizes with the TRAINED-LIKE weight set (round 6: what the driver runs must hold the bar where the headline is measured) ----
def _f64(sd):
    return {k: v.double() for k, v in sd.items()}


def test_config2_1024_trained_like_against_fp32_and_float64_oracle(weights_tl):
    """configs[1] (1024 x 1024 content, 512 x 512 style, alpha 0.5) with the weight statistics of the checkpoint the reference really
    loads (tests/test_gpu_trained_like.py): relative L2 <= 1e-4 against the fp32 oracle - the bar of every stage in both regimes;
    measured 7.6e-5, 86.7 dB - and no further from the float64 oracle than 4 x the fp32 oracle itself is (measured 3.0 x)."""
    from applied_image_processing_amd.engine import AdaINEngine

    vgg_sd, dec_sd = weights_tl
    c, s = T(synth.image(3, 1, 1024, 1024)), T(synth.image(4, 1, 512, 512))
    eng = AdaINEngine(vgg_sd, dec_sd, "cuda:0")
    eng.set_style(s.cuda())
    out = eng.stylize(c.cuda(), 0.5).cpu()
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5)
        tru = O.style_transfer_simple(_f64(vgg_sd), _f64(dec_sd), c.double(), s.double(), 0.5)
    rel = float((out - ref).norm() / ref.norm())
    mine, floor = float((out.double() - tru).norm() / tru.norm()), float((ref.double() - tru).norm() / tru.norm())
    psnr = float(O.psnr(out.clamp(0, 1), ref.clamp(0, 1)).min())
    print(f"config2 trained-like: relative L2 {rel:.3e} (float64: {mine:.3e}, the fp32 oracle itself {floor:.3e}), PSNR {psnr:.1f} dB")
    assert rel <= 1e-4 and psnr >= 60.0 and mine <= 4.0 * floor


def test_config4_1080p_frame_trained_like_uint8_one_call(weights_tl):
    """One 1080p video frame (configs[3]) through the ONE-CALL entry point (adain_stylize_u8: decoded uint8 in, finished uint8 out)
    with the trained-like set: at most one LSB from the quantised fp32 oracle, in at most 1.5 % of the bytes (measured 0.8 %), and
    the float image under the same 1e-4."""
    from applied_image_processing_amd.engine import AdaINEngine

    vgg_sd, dec_sd = weights_tl
    cu8 = (synth.image(7, 1, 1080, 1920)[0].transpose(1, 2, 0) * np.float32(255)).astype(np.uint8)
    c = T(cu8.transpose(2, 0, 1).copy()).float().div(255).unsqueeze(0)
    s = T(synth.image(4, 1, 512, 512))
    eng = AdaINEngine(vgg_sd, dec_sd, "cuda:0")
    eng.set_style(s.cuda())
    u8 = eng.stylize_u8(T(cu8[None]).cuda(), alpha=0.5).cpu()
    out = eng.stylize(T(cu8[None]).cuda(), 0.5).cpu()
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5)
    rel = float((out - ref).norm() / ref.norm())
    d = (u8.int() - O.quantize_u8(ref).int()).abs()
    off = float((d > 0).float().mean())
    print(f"config4 trained-like 1080p: relative L2 {rel:.3e}, uint8 max |diff| {int(d.max())}, {100 * off:.2f} % of the bytes off by one")
    assert tuple(u8.shape) == (1, 1080, 1920, 3) and rel <= 1e-4
    assert int(d.max()) <= 1 and off <= 0.015
    assert torch.equal(u8, eng.to_u8(out.cuda()).cpu())


def test_config3_depth_aware_2048(engine, weights):
    vgg_sd, dec_sd = weights
    c = T(synth.image(5, 1, 2048, 2048))
    s = T(synth.image(4, 1, 512, 512))
    d = T(synth.smooth_depth(6, 2048, 2048))
    out = engine.stylize_depth(c.cuda(), [d.cuda()], 0.15, 20)
    assert tuple(out.shape) == (1, 3, 2048, 2048)
    with torch.no_grad():
        ref = O.style_transfer(vgg_sd, dec_sd, c, s, d, 1.0, 0.15, 20)
    check(out, ref, "config3 2048^2 depth-aware")
    # the strength map itself
    from applied_image_processing_amd import runtime as rt

    p = rt.strength_map(d.cuda(), 256, 256, 0.15, 20).cpu()
    np.testing.assert_allclose(p.numpy(), O.compute_stylization_strength_map(d, (256, 256), 0.15, 20).numpy(), rtol=1e-4, atol=2e-5)
    assert float(p.max()) <= 0.85 + 1e-7


def test_config4_video_frames_1080p(engine, weights):
    vgg_sd, dec_sd = weights
    frames = T(np.concatenate([synth.image(7 + i, 1, 1080, 1920) for i in range(3)]))
    s = T(synth.image(4, 1, 512, 512))
    out = engine.stylize(frames.cuda(), 0.5)
    assert tuple(out.shape) == (3, 3, 1080, 1920)            # 1080 = 8 * 135
    with torch.no_grad():
        ref0 = O.style_transfer_simple(vgg_sd, dec_sd, frames[:1], s, 0.5)
    check(out[:1], ref0, "config4 1080p frame 0")
    for i in range(3):                                        # batch == frame by frame, bitwise
        assert torch.equal(engine.stylize(frames[i:i + 1].cuda(), 0.5)[0], out[i])
    u8 = engine.to_u8(out).cpu()
    assert tuple(u8.shape) == (3, 1080, 1920, 3) and u8.dtype == torch.uint8
    assert torch.equal(u8, O.quantize_u8(out.cpu()))          # same float input -> bit-exact quantiser
    # per-frame depth variant (reference video/utils.py:341-350 passes use_depth=True)
    d = T(synth.smooth_depth(9, 1080, 1920))
    outd = engine.stylize_depth(frames[:1].cuda(), [d.cuda()], 0.30, 20)
    with torch.no_grad():
        refd = O.style_transfer(vgg_sd, dec_sd, frames[:1], s, d, 1.0, 0.30, 20)
    check(outd, refd, "config4 1080p frame 0 depth-aware")


def test_config5_guide_views_1200x1600_masked(engine, weights, tmp_path):
    vgg_sd, dec_sd = weights
    s = T(synth.image(4, 1, 512, 512))
    views = []
    for i in range(2):
        v = synth.image(1000 + i, 1, 1200, 1600)
        bg = synth.uniform01(2000 + i, 1200 * 1600).reshape(1, 1, 1200, 1600) < 0.3    # ~30 % exact-zero background
        views.append(np.where(bg, np.float32(0), v))
    c = T(np.concatenate(views))
    mask = (c > 0)                                                                  # [n,3,H,W] bool (train.py:97 per view)
    out = engine.stylize(c.cuda(), 0.5)
    comp = engine.composite(c.cuda(), out, mask.float().cuda())
    assert tuple(comp.shape) == (2, 3, 1200, 1600)
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg_sd, dec_sd, c[:1], s, 0.5)
        refc = O.mask_composite(c[:1], ref, mask[0])
    check(comp[:1], refc, "config5 1200x1600 view 0 masked composite")
    assert torch.equal(comp[:, :, :][~mask].cpu(), c[~mask])                         # background pixels untouched
    # the batched guide writer keeps the reference's file naming (train.py:99-114)
    from PIL import Image
    from applied_image_processing_amd.engine import pooled_style_embedding, precompute_guides

    pil = [Image.fromarray((v[0].transpose(1, 2, 0) * 255).astype(np.uint8)) for v in views]
    masks = [np.asarray(p.resize((682, 512))).transpose(2, 0, 1) > 0 for p in pil]
    paths = precompute_guides(engine, pil, ["r_0", "r_1"], tmp_path / "stylized", masks=masks, content_size=512)
    assert sorted(paths) == ["r_0", "r_1"] and all(p.exists() and p.suffix == ".jpg" for p in paths.values())
    assert Image.open(paths["r_0"]).size == (682, 512)
    emb = pooled_style_embedding(engine.features(s.cuda()).permute(0, 3, 1, 2))
    with torch.no_grad():
        want = torch.nn.functional.adaptive_avg_pool2d(O.encode(vgg_sd, s), (1, 1)).view(1, 512)
    np.testing.assert_allclose(emb.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_run_depth_cli_offline(weights, tmp_path):
    from PIL import Image
    from applied_image_processing_amd.AdaIN import run_depth

    torch.save(synth.to_torch(synth.vgg_state_dict(0, full=True)), tmp_path / "vgg.pth")
    torch.save(synth.to_torch(synth.decoder_state_dict(0)), tmp_path / "dec.pth")
    Image.fromarray((synth.image(71, 1, 80, 96)[0].transpose(1, 2, 0) * 255).astype(np.uint8)).save(tmp_path / "c.png")
    Image.fromarray((synth.image(72, 1, 64, 64)[0].transpose(1, 2, 0) * 255).astype(np.uint8)).save(tmp_path / "s.png")
    np.save(tmp_path / "d.npy", synth.smooth_depth(73, 80, 96))
    p = run_depth.main(["--content", str(tmp_path / "c.png"), "--style", str(tmp_path / "s.png"), "--output", str(tmp_path / "o"),
                        "--use_depth", "--depth_npy", str(tmp_path / "d.npy"), "--vgg", str(tmp_path / "vgg.pth"),
                        "--decoder", str(tmp_path / "dec.pth")])
    assert p == tmp_path / "o" / "stylized.jpg" and p.exists()
    assert Image.open(p).size == (616, 512)      # content_size=512 default: 80x96 -> 512x614 -> decoder 8*ceil: 512x616


def test_graph_replay_is_bitwise_identical(engine):
    """The C ABI only enqueues work (no allocation, no synchronisation inside), so a whole stylize pass can be captured
    into a hipGraph; the replay must give exactly the eager result."""
    from applied_image_processing_amd.engine import GraphedStylize

    x = T(synth.image(201, 2, 96, 160)).cuda()
    g = GraphedStylize(engine, 2, 96, 160, alpha=0.5, to_u8=True)
    want = engine.to_u8(engine.stylize(x, 0.5))
    assert torch.equal(g(x), want)
    y = T(synth.image(202, 2, 96, 160)).cuda()
    assert torch.equal(g(y), engine.to_u8(engine.stylize(y, 0.5)))


def _hashf(tag, n):
    """numpy mirror of hashf() in examples/c_abi_smoke.c (uint32 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint32)
        x = (np.uint32(tag) * np.uint32(0x9E3779B1)) ^ (i * np.uint32(0x85EBCA77))
        x ^= x >> np.uint32(15); x *= np.uint32(0x2C1B3C6D)
        x ^= x >> np.uint32(12); x *= np.uint32(0x297A2D39)
        x ^= x >> np.uint32(15)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI: a plain-C program (no Python, no torch) linked against libadain_hip.so runs the whole
    path; its uint8 image must equal, bit for bit, what the Python surface produces from the same weights and inputs."""
    import os
    import re
    import shutil
    import subprocess

    from conftest import ROOT
    import applied_image_processing_amd.runtime as rt

    cc = shutil.which("cc") or shutil.which("gcc")
    if cc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no C compiler / ROCm headers on this box")
    pkg = os.path.dirname(rt.LIB_PATH)
    exe = tmp_path / "c_abi_smoke"
    subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "examples", "c_abi_smoke.c"), "-o", str(exe), "-L", pkg, "-ladain_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-lm"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=pkg + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    H, W, Hs, Ws = 72, 104, 64, 80
    r = subprocess.run([str(exe), str(H), str(W), str(Hs), str(Ws)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.search(r"-> (\d+)x(\d+) image, sum (\d+), fnv ([0-9a-f]{16})", r.stdout)
    assert m, r.stdout

    # the same weights / inputs through the Python surface
    enc = [(3, 3, 1), (3, 64, 3), (64, 64, 3), (64, 128, 3), (128, 128, 3), (128, 256, 3), (256, 256, 3), (256, 256, 3),
           (256, 256, 3), (256, 512, 3)]
    dec = [(512, 256, 3), (256, 256, 3), (256, 256, 3), (256, 256, 3), (256, 128, 3), (128, 128, 3), (128, 64, 3),
           (64, 64, 3), (64, 3, 3)]
    vgg_sd, dec_sd = {}, {}
    for i, ((cin, cout, k), key) in enumerate(zip(enc, rt.ENC_KEYS)):
        bound = np.sqrt(np.float32(6.0) / np.float32(cin * k * k), dtype=np.float32)
        vgg_sd[f"{key}.weight"] = T((_hashf(100 + i, cout * cin * k * k) * (np.float32(2.0) * bound)).reshape(cout, cin, k, k))
        vgg_sd[f"{key}.bias"] = T(_hashf(200 + i, cout) * np.float32(0.1))
    for i, ((cin, cout, k), key) in enumerate(zip(dec, rt.DEC_KEYS)):
        bound = np.sqrt(np.float32(6.0) / np.float32(cin * 9), dtype=np.float32)
        dec_sd[f"{key}.weight"] = T((_hashf(300 + i, cout * cin * 9) * (np.float32(2.0) * bound)).reshape(cout, cin, 3, 3))
        dec_sd[f"{key}.bias"] = T(_hashf(400 + i, cout) * np.float32(0.1))
    from applied_image_processing_amd.engine import AdaINEngine

    e = AdaINEngine(vgg_sd, dec_sd, "cuda:0")
    content = T((_hashf(1, 3 * H * W) + np.float32(0.5)).reshape(1, 3, H, W)).cuda()
    style = T((_hashf(2, 3 * Hs * Ws) + np.float32(0.5)).reshape(1, 3, Hs, Ws)).cuda()
    e.set_style(style)
    u8 = e.to_u8(e.stylize(content, 0.5)).cpu().numpy().reshape(-1)
    assert (int(m.group(1)), int(m.group(2))) == (72, 104)
    assert int(m.group(3)) == int(u8.astype(np.uint64).sum())
    fnv = 1469598103934665603
    for b in u8.tolist():
        fnv = ((fnv ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert m.group(4) == f"{fnv:016x}"
    # ... and the one-call entry point on the same frame as decoded bytes with a mask
    m2 = re.search(r"adain_stylize_u8 masked frame -> (\d+)x(\d+), sum (\d+), fnv ([0-9a-f]{16})", r.stdout)
    assert m2, r.stdout
    frame = ((_hashf(1, 3 * H * W) + np.float32(0.5)) * np.float32(255.0)).astype(np.uint8).reshape(3, H, W)
    fu8 = T(frame.transpose(1, 2, 0)).cuda().unsqueeze(0)
    got = e.stylize_u8(fu8, alpha=0.5, masks=T(frame > 96).cuda().unsqueeze(0)).cpu().numpy().reshape(-1)
    assert (int(m2.group(1)), int(m2.group(2))) == (H, W) and int(m2.group(3)) == int(got.astype(np.uint64).sum())
    fnv = 1469598103934665603
    for b in got.tolist():
        fnv = ((fnv ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert m2.group(4) == f"{fnv:016x}"


def _bench(args, timeout=240):
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_front_door_starts_its_own_ranks():
    """``python bench.py --gpus 2`` typed bare, as the driver types it: two fresh rank processes (here they share the one GPU:
    ``--rehearse``, labelled), ONE JSON line from rank 0, exit status 0; the default gather mode is the end-of-region gather and
    the line shows what the transport saw.  Without ``--rehearse`` two ranks on one GPU are refused with a clear message."""
    import json

    r = _bench(["--gpus", "2", "--rehearse", "--no-cpu", "--no-secondary", "--steps", "3", "--warmup", "1", "--size", "256"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rehearsal"]["ranks_share_a_gpu"]
    assert d["gather"]["mode"] == "end" and d["gather"]["gathers_in_timed_region"] == 1
    assert d["ranks"]["world"] == 2 and d["ranks"]["allreduce_of_ones"] == 2 and len(d["ranks"]["devices"]) == 2
    assert d["ranks"]["launcher"] == "bench.py self-launch" and len({x["pid"] for x in d["ranks"]["devices"]}) == 2
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["compute_ms"] > 0 for p in d["per_rank"])
    for p in d["per_rank"]:            # round 6: each rank's own GPU clock and power over its timed region (sysfs, side thread)
        g = p["gpu"]
        if not g["available"]:         # a box that hides the GPU's hwmon node from the container: the line must say why, nothing else breaks
            assert g["why"], g
            continue
        assert g["power_cap_w"] is None or g["power_cap_w"] > 100, g
        assert g["timed_region"]["samples"] >= 1 and 100 <= g["timed_region"]["sclk_mhz"]["median"] <= 3000, g
        assert g["timed_region"]["power_w"]["median"] > 50, g
    if __import__("torch").cuda.device_count() < 2:
        r = _bench(["--gpus", "2", "--no-cpu", "--steps", "3"], timeout=120)
        assert r.returncode != 0 and "2 ranks, one per GPU" in (r.stderr + r.stdout)
