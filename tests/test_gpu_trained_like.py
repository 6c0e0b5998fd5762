"""GPU parity at TRAINED-LIKE weight statistics (round 5).  The checkpoint the reference really loads
(test.py:183-185; an LFS pointer in the reference tree) has a conv0 with Caffe-style preprocessing (x255, channel swap, biases near
-104 ... -124), filters with non-zero means, zero-sum edge detectors and channels normalised to a post-ReLU mean of 1 - magnitudes
and cancellation no Kaiming-weight test reaches.  ``synth.trained_like_state_dicts`` rebuilds that regime from the integer PRNG;
``tests/golden/case_g.npz`` holds what the UNMODIFIED reference computes with it (fp32, and float64 as the yardstick).  Here: the
folded first layer (conv0 into conv1_1, float and uint8 entry), the whole F(4,3) x F(2,3) network, statistics, AdaIN, alpha and
depth-aware outputs, through the C ABI.  Tolerance and measurements: see TOL / FLOOR below.
Run with ``-m gpu``."""
import json
import os

import numpy as np
import pytest
import torch

import applied_image_processing_amd.synth as synth
from conftest import ROOT, golden
from oracle import adain_oracle as O
from test_oracle_golden import g_inputs, rel_l2

pytestmark = pytest.mark.gpu

REPORT = {}
# The bar, the same in every file that states one (README, DESIGN section 2, tests/fuzz_shapes.py, tests/verify_real_weights.py --tol):
# relative L2 <= 1e-4 against the reference's fp32 output at EVERY stage, in both weight regimes.  This file additionally holds each
# stage of the trained-like regime at 1.5 x what it measures today, and at a multiple of the reference's OWN fp32 rounding noise (its
# distance from its float64 run: case_g holds both), so that a regression of an intermediate stage cannot hide under the image bound.
# History, said plainly: round 5 first stated 1e-4 here, the first run read 1.09e-4 on `out_a10`, and the tolerance was raised to a
# blanket 2e-4; the same round then packed U = G g G^T from double (a systematic few-ulp error per weight gone) - and nobody re-read
# the numbers: on that binary and since, every image is at 6.9 - 8.0e-5 (gpurun_out/trained_like_report.json ->
# profiles/r06_trained_like_parity.json).  Round 6 therefore states 1e-4 again and removes the 2e-4.
# What is left of the factor ~2.8 over the reference's own noise is the fp32 accumulation over cin in the Winograd domain: per layer
# 7e-7 at 64 channels per chain, 1.25e-6 at 256 (profiles/r06_wino_error_per_layer_gpu.txt: chains cut to 64 channels - the latency
# schedule's cin split - bring every layer back to 6.5e-7).
TOL = 1e-4
#              rel. L2 vs reference fp32 (measured)    x the reference's own distance from float64 (measured)
BOUNDS = {"relu1_1": (1e-6,   3.0),                 # 2.5e-7                                   2.0
          "content_f": (2e-5, 3.5), "style_f": (2e-5, 3.5),     # 1.1 - 1.3e-5                 2.7 - 2.9
          "mean": (5e-6, 3.5), "std": (7e-6, 3.5),              # 2.7 - 3.1e-6, 3.9 - 4.5e-6   2.4 - 2.8
          "adain": (2e-5, 3.5),                                 # 1.3e-5                       2.5 - 2.9
          "out_a05": (TOL, 4.0), "out_a10": (TOL, 4.0), "out_depth": (TOL, 4.0)}      # 6.9 - 8.0e-5   2.5 - 3.1
FLOOR = 4.0          # larger frames against the oracle: 3.0 - 3.1 measured


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous().cpu().numpy()


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import applied_image_processing_amd.runtime as rt

    rt.lib()
    return rt


@pytest.fixture(scope="module")
def nets_tl(weights_tl):
    """The reference-shaped singletons with the trained-like set loaded (Kaiming weights again afterwards: other modules load their own)."""
    from applied_image_processing_amd.AdaIN import net

    net.vgg.load_state_dict(weights_tl[0])
    net.decoder.load_state_dict(weights_tl[1])
    net.vgg.to("cuda").eval()
    net.decoder.to("cuda").eval()
    yield net.vgg, net.decoder
    net.vgg.load_state_dict(synth.to_torch(synth.vgg_state_dict(0, full=True)))
    net.decoder.load_state_dict(synth.to_torch(synth.decoder_state_dict(0)))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and REPORT:
        with open(os.path.join(out, "trained_like_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)


def check(tag, key, got, g):
    """``got`` against the reference's fp32 output (<= the stage's bound) and against its float64 run (<= the stage's multiple of the
    reference's own distance from it)."""
    tol, floor_factor = BOUNDS[key]
    ref, f64 = g[f"{tag}_{key}"], g[f"{tag}_{key}_f64"]
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    assert got.shape == ref.shape, (got.shape, ref.shape)
    r, mine, floor = rel_l2(got, ref), rel_l2(got, f64), rel_l2(ref, f64)
    REPORT[f"{tag}_{key}"] = {"gpu_vs_reference_fp32": r, "gpu_vs_reference_f64": mine, "reference_fp32_vs_f64": floor}
    assert r <= tol, (tag, key, r)
    assert mine <= floor_factor * floor + 1e-7, (tag, key, mine, floor)


@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_folded_first_layer_at_caffe_magnitudes(rt, weights_tl, tag):
    """vgg[:4] of the reference = conv0 (x255, swap, -104 ... -124) -> pad -> conv1_1 -> relu; here ONE layer with conv0 folded into
    conv1_1's weights and bias (csrc/conv_edge.hip:pack_conv_first_kernel): the x255 products cancel against a folded bias of order
    100 x sum|w|.  Float entry and uint8 entry (ToTensor inside the kernel) must agree bit for bit."""
    g = golden("case_g.npz")
    cu8, c, _, _ = g_inputs(tag)
    packed = rt.pack_encoder(weights_tl[0], torch.device("cuda:0"))
    a = rt.encode_relu1_1(c.cuda(), packed)
    b = rt.encode_relu1_1(T(cu8[None]).cuda(), packed)
    assert torch.equal(a, b)
    got = nchw(a)
    check(tag, "relu1_1", got, g)          # measured 2.5e-7: 2 x the unfolded reference's own 1.1e-7 from float64
    assert float(np.abs(got - g[f"{tag}_relu1_1"]).max()) <= 1e-4 * float(np.abs(g[f"{tag}_relu1_1"]).max())


@pytest.mark.parametrize("tag", ["sq", "odd"])
def test_whole_network_against_case_g(rt, nets_tl, tag):
    from applied_image_processing_amd.AdaIN import function as fn, test as t

    vgg, dec = nets_tl
    g = golden("case_g.npz")
    cu8, c, s, depth = g_inputs(tag)
    c, s, depth = c.cuda(), s.cuda(), depth.cuda()
    cf, sf = vgg(c), vgg(s)
    check(tag, "content_f", cf, g)
    check(tag, "style_f", sf, g)
    m, sd = fn.calc_mean_std(cf)
    check(tag, "mean", m, g)
    check(tag, "std", sd, g)
    check(tag, "adain", fn.adaptive_instance_normalization(cf, sf), g)
    check(tag, "out_a05", t.style_transfer_simple(vgg, dec, c, s, 0.5), g)
    check(tag, "out_a10", t.style_transfer_simple(vgg, dec, c, s, 1.0), g)
    check(tag, "out_depth", t.style_transfer(vgg, dec, c, s, depth, 1.0, 0.15, 20), g)
    # the uint8 entry of the whole encoder on the same frame: bit-identical features
    packed = rt.pack_encoder({k: v for k, v in vgg.state_dict().items()}, torch.device("cuda:0"))
    assert torch.equal(rt.encode_u8(T(cu8[None]).cuda(), packed), rt.encode(c, packed))


@pytest.mark.parametrize("h,w,hs,ws", [(256, 256, 128, 160), (200, 328, 96, 96)])
def test_larger_frames_against_the_oracle_and_float64(rt, weights_tl, h, w, hs, ws):
    """Beyond the fixture sizes (persistent tile lists, both tile geometries, the one-call uint8 entry point): the oracle in fp32 is
    the reference's arithmetic, the oracle in float64 the yardstick."""
    from applied_image_processing_amd.engine import AdaINEngine

    vgg, dec = weights_tl
    cu8 = (synth.image(71, 1, h, w)[0].transpose(1, 2, 0) * np.float32(255)).astype(np.uint8)
    c = T(cu8.transpose(2, 0, 1).copy()).float().div(255).unsqueeze(0)
    s = T(synth.image(72, 1, hs, ws))
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg, dec, c, s, 0.5)
        v64, d64 = {k: v.double() for k, v in vgg.items()}, {k: v.double() for k, v in dec.items()}
        f64 = O.style_transfer_simple(v64, d64, c.double(), s.double(), 0.5).float()
    eng = AdaINEngine(vgg, dec, "cuda:0")
    eng.set_style(s.cuda())
    got = eng.stylize(c.cuda(), 0.5).cpu()
    r, mine, floor = rel_l2(got, ref.numpy()), rel_l2(got, f64.numpy()), rel_l2(ref, f64.numpy())
    REPORT[f"oracle_{h}x{w}"] = {"gpu_vs_oracle_fp32": r, "gpu_vs_oracle_f64": mine, "oracle_fp32_vs_f64": floor}
    assert r <= TOL and mine <= FLOOR * floor, (r, mine, floor)
    u8 = eng.stylize_u8(T(cu8[None]).cuda(), alpha=0.5).cpu()
    d = (u8.int() - O.quantize_u8(ref).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-2          # measured: 0.8 % of the bytes one LSB off
    assert torch.equal(u8, eng.to_u8(eng.stylize(T(cu8[None]).cuda(), 0.5)).cpu())


def test_real_checkpoint_acceptance_script_runs_end_to_end(rt, weights_tl, tmp_path, capsys):
    """tests/verify_real_weights.py, compute part: checkpoints in the reference's key layout (the trained-like set standing in for
    the files nobody here has), the reference's sample pair at 256 from tests/golden/case_f.npz; passes at this regime's stated
    tolerance, fails (non-zero, the stage named) at an impossible one."""
    import verify_real_weights as v

    torch.save(weights_tl[0], tmp_path / "vgg.pth")
    torch.save(weights_tl[1], tmp_path / "dec.pth")
    base = ["--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "dec.pth"), "--content", "", "--style", "", "--any-weights", "--sizes", "256", "512"]
    rc = v.main(base + ["--tol", str(TOL)])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 0 and rep["verdict"] == "PASS", rep["problems"]
    st = rep["runs"]["256"]["stages"]
    assert st["uint8 image"]["max_abs_lsb"] <= 1 and st["output"]["psnr_db"] > 60 and "skipped" in rep["runs"]["512"]
    assert rep["weights"]["conv0_bias"] == [-103.939, -116.779, -123.68]
    REPORT["real_image_pair_256"] = st
    rc = v.main(base + ["--tol", "1e-9"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 1 and rep["verdict"] == "FAIL" and any("relu4_1(content)" in p for p in rep["problems"])
