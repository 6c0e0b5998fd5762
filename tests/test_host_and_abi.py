"""CPU-only tests: the C-ABI library loads and exports every symbol include/adain_hip.h declares, host
logic (architecture tables, transforms, sharding), and that the product path fails loudly without a GPU
or without the library (no silent fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

import applied_image_processing_amd.arch as arch
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.sharding as sh
import applied_image_processing_amd.synth as synth


def _lib_built():
    if not os.path.exists(rt.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return rt.lib()


def _declared(header_name):
    header = open(os.path.join(ROOT, "include", header_name)).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(adain_[a-z0-9_]+)\s*\(", header)))


def test_schedule_is_per_thread_state():
    """adain_set_schedule: enqueue-time state of the calling thread (include/adain_hip.h) - returns the previous value, refuses unknown
    values, another thread keeps the default, the context manager restores."""
    import threading

    lib = _lib_built()
    assert rt.get_schedule() == rt.SCHEDULE_BATCH
    assert rt.set_schedule(rt.SCHEDULE_LATENCY) == rt.SCHEDULE_BATCH and rt.get_schedule() == rt.SCHEDULE_LATENCY
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.adain_get_schedule()))
    t.start(); t.join()
    assert seen == [rt.SCHEDULE_BATCH]
    assert lib.adain_set_schedule(7) == -1 and b"unknown schedule" in lib.adain_last_error()
    assert rt.get_schedule() == rt.SCHEDULE_LATENCY
    assert rt.set_schedule(rt.SCHEDULE_BATCH) == rt.SCHEDULE_LATENCY
    with rt.schedule(rt.SCHEDULE_LATENCY):
        assert rt.get_schedule() == rt.SCHEDULE_LATENCY
    assert rt.get_schedule() == rt.SCHEDULE_BATCH
    # without a device there is nothing to split for: the slab query is 0 and the workspaces keep their two ping-pong buffers
    assert rt.conv3x3_wino4_split_bytes(1, 32, 57, 512, 256) == 0 or torch.cuda.is_available()


def _exported_functions(path):
    """Names of the defined dynamic symbols of type T (code) of a shared library."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T")


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared("adain_hip.h")
    assert len(declared) >= 25
    lib = _lib_built()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in adain_hip.h but not exported"
        assert name in rt.SIGNATURES, f"{name} has no ctypes signature in runtime.py"
    assert sorted(rt.SIGNATURES) == declared
    # -fvisibility=hidden: the C ABI is ALL the code the product library exports (no C++ launchers, no helper functions)
    assert _exported_functions(rt.LIB_PATH) == declared
    assert not hasattr(lib, "adain_conv3x3") and not hasattr(lib, "adain_debug_set_conv_stamp_buffer")
    assert lib.adain_abi_version() == rt.ABI_VERSION == 4
    assert lib.adain_encoder_packed_floats() > 3_500_000 and lib.adain_decoder_packed_floats() > 3_500_000
    hc, wc = ctypes.c_int(), ctypes.c_int()
    lib.adain_encoded_size(45, 67, ctypes.byref(hc), ctypes.byref(wc))
    assert (hc.value, wc.value) == (6, 9) == arch.encoded_size(45, 67)
    assert lib.adain_encode_workspace_bytes(1, 1024, 1024) == (64 + 16) * 1024 * 1024 * 4   # A: conv1_1 out, B: pooled conv1_2 out


def test_diagnostic_library_exports_both_headers(diag_lib):
    both = sorted(set(_declared("adain_hip.h")) | set(_declared("adain_hip_diag.h")))
    assert _exported_functions(diag_lib.LIB_PATH) == both
    assert sorted(set(diag_lib.SIGNATURES) | set(diag_lib.DIAG_SIGNATURES)) == both
    assert diag_lib.is_diag() and diag_lib.lib().adain_abi_version() == 4


def test_product_library_ignores_the_environment(monkeypatch):
    """ADAIN_HIP_LIB (round 2's switch) no longer redirects the product runtime; the retired kernel families say so."""
    import importlib

    monkeypatch.setenv("ADAIN_HIP_LIB", "/nonexistent/libother.so")
    fresh = importlib.reload(rt)
    try:
        assert fresh.LIB_PATH.endswith("libadain_hip.so") and not fresh.is_diag()
        with pytest.raises(fresh.AdainHipError, match="retired"):
            fresh.conv3x3_wino_pack(torch.zeros(64, 64, 3, 3), 3)
        assert not hasattr(fresh, "conv3x3_pack") and not hasattr(fresh, "conv3x3")
    finally:
        importlib.reload(rt)


def test_no_cpu_fallback():
    _lib_built()
    with pytest.raises(rt.AdainHipError):
        rt.encode(torch.zeros(1, 3, 32, 32), torch.zeros(8))
    with pytest.raises(rt.AdainHipError):
        rt.mean_std(torch.zeros(1, 4, 2, 2), False)
    if not torch.cuda.is_available():
        from applied_image_processing_amd.AdaIN import function as fn, test as t

        with pytest.raises(rt.AdainHipError):
            t._device()
        with pytest.raises(rt.AdainHipError):
            fn.calc_mean_std(torch.zeros(1, 4, 2, 2))
        with pytest.raises(rt.AdainHipError):
            t.compute_stylization_strength_map(torch.zeros(4, 4), (2, 2))


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(rt, "_lib", None)
    monkeypatch.setattr(rt, "LIB_PATH", "/nonexistent/libadain_hip.so")
    with pytest.raises(rt.AdainHipError, match="missing"):
        rt.lib()


def test_arch_tables_match_reference_layout():
    assert arch.conv_indices(arch.VGG_MODULES[: arch.ENCODER_CUT]) == rt.ENC_KEYS
    assert arch.conv_indices(arch.DECODER_MODULES) == rt.DEC_KEYS
    from applied_image_processing_amd.AdaIN import net

    vsd, dsd = net.vgg.state_dict(), net.decoder.state_dict()
    # 34 tensors / 80,097,584 B and 18 tensors / 14,020,876 B
    assert len(vsd) == 34 and sum(v.numel() * 4 for v in vsd.values()) == 80_097_584
    assert len(dsd) == 18 and sum(v.numel() * 4 for v in dsd.values()) == 14_020_876
    assert set(synth.vgg_state_dict(0, full=True)) == set(vsd)
    assert set(synth.decoder_state_dict(0)) == set(dsd)
    net.vgg.load_state_dict(synth.to_torch(synth.vgg_state_dict(0, full=True)))     # strict
    net.decoder.load_state_dict(synth.to_torch(synth.decoder_state_dict(0)))
    # work model of SURVEY.md section 8(d)
    assert arch.conv_flops_encoder(1024, 1024) == 482_706 * 1024 * 1024
    assert arch.conv_flops_decoder(128, 128) == 482_688 * 1024 * 1024
    plan = arch.encoder_plan()
    assert [p["src"] for p in plan] == ["direct", "direct", "direct", "pool", "direct", "pool", "direct", "direct", "direct", "pool"]
    assert [p["src"] for p in arch.decoder_plan()] == ["direct", "up", "direct", "direct", "direct", "up", "direct", "up", "direct"]


def test_reference_state_dict_keys_if_reference_present():
    from oracle import ref_loader

    if not ref_loader.available():
        pytest.skip("reference tree not present")
    _, net_ref, _ = ref_loader.load()
    from applied_image_processing_amd.AdaIN import net

    for mine, ref in ((net.vgg, net_ref.vgg), (net.decoder, net_ref.decoder)):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape for k in a)
        assert [type(m).__name__ for m in mine.children()] == [type(m).__name__ for m in ref.children()]


def test_test_transform_semantics():
    from PIL import Image
    from applied_image_processing_amd.AdaIN import test as t

    assert t._resize_size(933, 700, 256) == (341, 256)        # W x H, SURVEY.md 8(c)
    assert t._resize_size(512, 512, 256) == (256, 256)
    assert t._resize_size(1600, 1200, 512) == (682, 512)
    img = Image.fromarray((synth.image(5, 1, 40, 60)[0].transpose(1, 2, 0) * 255).astype(np.uint8))
    x = t.test_transform(0, False)(img)
    assert x.shape == (3, 40, 60) and x.dtype == torch.float32
    assert torch.equal(x, torch.from_numpy(np.asarray(img).transpose(2, 0, 1).copy()).float() / 255)
    y = t.test_transform(20, True)(img)
    assert y.shape == (3, 20, 20)
    z = t.test_transform(20, False)(img)
    assert z.shape == (3, 20, 30)
    rgba = img.convert("RGBA")
    assert t.test_transform(0, False)(rgba).shape == (4, 40, 60)
    gray = img.convert("L")
    assert t.test_transform(0, False)(gray).shape == (1, 40, 60)


def test_style_object_fingerprint_moves_with_in_place_edits():
    """The style cache keys a PIL object by identity AND by a fingerprint of its pixels (AdaIN/test.py:_pixel_fingerprint): any
    in-place edit - one pixel, a swap of two pixels, a paste - must move it; an untouched image keeps it."""
    from PIL import Image
    from applied_image_processing_amd.AdaIN import test as t

    base = (synth.image(6, 1, 70, 93, c=4)[0].transpose(1, 2, 0) * 255).astype(np.uint8)
    for mode, a in (("RGB", base[..., :3]), ("RGBA", base), ("L", base[..., 0])):
        img = Image.fromarray(np.ascontiguousarray(a), mode)
        f0 = t._pixel_fingerprint(img)
        assert f0 is not None and f0 == t._pixel_fingerprint(img) == t._pixel_fingerprint(img.copy())
        p, q = img.getpixel((3, 4)), img.getpixel((60, 50))
        assert p != q
        img.putpixel((3, 4), q)
        f1 = t._pixel_fingerprint(img)
        img.putpixel((60, 50), p)                         # now the two pixels are swapped: same histogram, another image
        f2 = t._pixel_fingerprint(img)
        img.paste(img.crop((0, 0, 20, 20)), (40, 30))
        assert len({f0, f1, f2, t._pixel_fingerprint(img)}) == 4, mode


def test_shard_ranges():
    assert sh.shard_counts(300, 8) == [38, 38, 38, 38, 37, 37, 37, 37]
    assert sh.shard_counts(512, 8) == [64] * 8
    assert sh.shard_counts(3, 8) == [1, 1, 1, 0, 0, 0, 0, 0]
    for n in (0, 1, 7, 300, 512):
        for w in (1, 2, 3, 8):
            spans = [sh.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    with pytest.raises(ValueError):
        sh.shard_range(4, 2, 2)


def test_device_side_generator_has_the_host_generators_bits():
    """synth.uniform01_torch / frame_u8_torch (bench.py --job builds its 512 frames with them on the GPU) against the numpy generator."""
    for seed in (0, 7, 1000, 2 ** 31 + 5):
        assert np.array_equal(synth.uniform01(seed, 4099), synth.uniform01_torch(seed, 4099).numpy())
    f = synth.frame_u8_torch(11, 9, 14).numpy()
    assert np.array_equal(f, np.floor(synth.image(11, 1, 9, 14)[0] * 256).astype(np.uint8).transpose(1, 2, 0))


def test_sharding_helpers_without_a_process_group():
    assert sh.chunk_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)] and sh.chunk_bounds(2, 3) == [(0, 1), (1, 2), (2, 2)]
    assert sh.agree(True) and not sh.agree(False) and sh.agree_min(1) == 1
    assert sh.agree_geometry(True, {(4, 6, 3)}) == (True, (4, 6, 3), True)
    assert sh.agree_geometry(True, {(4, 6, 3), (4, 8, 3)})[2] is False
    assert sh.agree_geometry(False, set()) == (False, None, True)


def test_bench_front_door_refuses_more_ranks_than_gpus_before_any_work():
    """``python bench.py --gpus N`` (no torchrun typed) on a box with fewer than N GPUs: a clear message and a non-zero status from the
    launcher itself, before any rank is started (here: no GPU at all)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2 means 2 ranks, one per GPU" in r.stderr and "{" not in r.stdout


def test_bench_launcher_stops_every_rank_when_one_fails_or_hangs(tmp_path):
    """``bench.self_launch`` with stand-in rank programs: rank 0 fails after a second while rank 1 would run for a minute -> the
    launcher reports rank 0's status, stops rank 1 and exits with that status within seconds; ranks that never finish hit the time
    limit (exit status 124).  The stand-ins also check the environment a rank is started with."""
    import subprocess
    import sys
    import time

    rank_prog = tmp_path / "rank.py"
    rank_prog.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "assert os.environ['WORLD_SIZE'] == '2' and os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert os.environ['MASTER_PORT'].isdigit() and os.environ['ADAIN_SELF_LAUNCHED'] == str(os.getppid())\n"
        "open(os.path.join(os.path.dirname(__file__), f'started_{r}'), 'w').close()\n"
        "if '--hang' not in sys.argv and r == 0:\n"
        "    time.sleep(1.0); sys.exit(3)\n"
        "time.sleep(60)\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        f"bench.__file__ = {str(rank_prog)!r}\n"
        "bench.self_launch()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, str(driver), "--gpus", "2", "--rehearse"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "rank 0 exited with status 3" in r.stderr, (r.returncode, r.stderr[-500:])
    assert (tmp_path / "started_0").exists() and (tmp_path / "started_1").exists() and time.time() - t0 < 40
    t0 = time.time()
    r = subprocess.run([sys.executable, str(driver), "--gpus", "2", "--rehearse", "--hang", "--launch-timeout", "3"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 124 and "did not finish within 3 s" in r.stderr and time.time() - t0 < 40, (r.returncode, r.stderr[-500:])


@pytest.mark.parametrize("header", ["adain_hip.h", "adain_hip_diag.h"])
def test_headers_are_plain_c_and_cxx(tmp_path, header):
    """The boundary is a C ABI: both headers compile on their own as strict C99 and as C++17 (no torch, no HIP types in any signature)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no gcc / g++ here")
    src = tmp_path / "use.c"
    src.write_text(f'#include "{header}"\nint main(void) {{ return adain_abi_version() == 0; }}\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Werror", "-fsyntax-only"], ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++"]):
        r = subprocess.run(cmd + ["-I", inc, str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_real_checkpoint_acceptance_script_checks_files_and_layout(tmp_path, capsys):
    """tests/verify_real_weights.py on synthetic ``.pth`` files of the reference key layout: the layout passes, the files are
    (correctly) reported as NOT the reference's checkpoints - size and sha256 differ from the LFS pointers - and a checkpoint with a
    missing / misshapen tensor is named.  The compute part needs a GPU (test_gpu_configs.py runs it there)."""
    import json

    import verify_real_weights as v

    vgg, dec = synth.to_torch(synth.vgg_state_dict(0, full=True)), synth.to_torch(synth.decoder_state_dict(0))
    torch.save(vgg, tmp_path / "vgg.pth")
    torch.save(dec, tmp_path / "dec.pth")
    assert v.check_layout(vgg, "vgg") == [] and v.check_layout(dec, "decoder") == []
    f = v.check_file(tmp_path / "vgg.pth", "vgg")
    assert not f["ok"] and not f["size_matches"] and f["sha256_matches"] is False and len(f["sha256"]) == 64
    assert v.POINTERS["vgg"]["size"] == 80102481 and v.POINTERS["decoder"]["size"] == 14023458       # the reference's LFS pointers
    rc = v.main(["--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "dec.pth"), "--files-only"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 1 and rep["verdict"] == "FAIL" and len(rep["problems"]) == 2 and all("not the reference's" in p for p in rep["problems"])
    rc = v.main(["--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "dec.pth"), "--files-only", "--any-weights"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 0 and rep["verdict"] == "PASS" and rep["weights"]["conv0_weight"][0] != 0
    broken = dict(dec)
    del broken["28.bias"]
    broken["1.weight"] = broken["1.weight"][:, :100]
    torch.save(broken, tmp_path / "bad.pth")
    rc = v.main(["--vgg", str(tmp_path / "vgg.pth"), "--decoder", str(tmp_path / "bad.pth"), "--files-only", "--any-weights"])
    rep = json.loads(capsys.readouterr().out)
    assert rc == 1 and any("missing key 28.bias" in p for p in rep["problems"]) and any("1.weight: shape" in p for p in rep["problems"])
