#!/usr/bin/env python3
"""Throughput benchmark of the AdaIN hot path on MI355X (BASELINE.json metric: stylised Mpixels/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]
N > 1: one process per GPU.  Either launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``, or
typed bare (``python bench.py --gpus N``): the bare process then starts N fresh rank processes itself BEFORE anything touches a
GPU (subprocesses, never an exec), hands rank 0's JSON line through, and exits with the worst child's status.

A "step" is one pass of the hot path over one batch of synthetic frames that are resident in HBM when the timed region starts:
  config 2 (default, BASELINE configs[1])  one full ``style_transfer_simple`` (reference Style_3DGS/AdaIN/test.py:74-81): encode the
            1024x1024 content, encode the 512x512 style (re-encoded every step, as every reference call does), statistics,
            AdaIN + alpha blend, decode.
  config 3  the depth-aware ``style_transfer`` (test.py:52-71) at 2048x2048 with a proximity map.
  config 4  one sub-batch of the video job (video/utils.py:297-369): ``--batch`` 1080p frames per GPU per step (one style for the
            clip: its statistics are computed once per rank), uint8 out.  ``--job`` runs the whole 512-frame job (below).
  config 5  one sub-batch of the 3DGS guide-view job (Style_3DGS/train.py:86-115): ``--batch`` masked 1200x1600 views per GPU per
            step, mask composite + uint8 out.  ``--job`` runs the whole 300-view job.
With more than one rank (or under torch.distributed.run) the frame list of a step is cut into contiguous per-rank blocks
(weak scaling: ``--batch`` frames per GPU) and the finished uint8 frames meet on rank 0 in the path's one collective, over
RCCL: ``--gather end`` (default) keeps every rank's frames of the K timed steps in HBM and ENDS the timed region with ONE gather
of them (the region is one K x N-frame job; no transport kernel ever runs beside the compute kernels); ``--gather overlap``
issues one asynchronous gather per step (step k's overlaps step k+1's compute).  Both are inside the timing.  The
device gather must run over RCCL ("nccl" backend): if the process group cannot provide it, or if there are more ranks than
GPUs, the benchmark exits non-zero instead of silently degrading (``--rehearse`` allows both for single-GPU rehearsals and
labels the JSON line).  The timed region is bracketed by barrier + synchronize; the MAX over ranks is reported.

``--job`` (configs 4 / 5) runs the BASELINE job itself instead of one frame per GPU per step: 512 frames / 300 views
(``--frames``) cut into contiguous blocks over the ranks (strong scaling: 64 per GPU at 8; 38 / 37 ragged), every rank's block
resident in HBM as decoded uint8 frames, ``jobs.stylize_frames_sharded`` with ``--batch`` frames per sub-batch (no host
collective and no device synchronisation inside the frame loop, one status word and ONE gather per job); a "step" is one whole
job, ``value`` = job pixels / wall time barrier to barrier including the gather.  ``--host-frames`` keeps the frames in
(pageable) host memory instead: the job driver's feeder stages them in pinned buffers and uploads them on a copy stream behind
the kernels, and the gathered result is copied back to the host - the PCIe-inclusive rate of the same job, reported as
``pcie_inclusive`` beside an HBM-resident ``value`` measured in the same run.

Rank 0 prints ONE JSON line.
  roofline      the dominant kernel family (3x3 convolutions, fp32 MFMA): HIP events recorded by the C ABI on the launch stream
                around every conv launch.  ``achieved`` = multiplies the kernel EXECUTES on the matrix pipe (the Winograd
                F(4,3) x F(2,3) form needs 24 per 72 of the direct algorithm's) x 2 flop / time, ``frac`` = achieved / 157.3
                TFLOP/s (<= 1 by construction); the direct-convolution rate is reported as ``algorithmic_tflops``.
  secondary     the HBM-bound kernels: algorithmic bytes / event time, against 8 TB/s.
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference path) timed on this node's host cores on a bounded
                sample of the same workload; it also yields the PSNR / relative L2 of the GPU output.
"""
import argparse
import glob
import json
import math
import os
import statistics
import sys
import time

_T0 = time.perf_counter()          # this process's start, for `run_phases_s`
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
# this pool's host driver only supports dmabuf IPC: without this RCCL's communicator set-up between processes fails with
# `hipIpcGetMemHandle: invalid argument`.  The launch environment exports it; kept here for a bare shell (it must be set before the
# first HIP call of the process, i.e. before anything touches the GPU)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")



def _argv_value(flag, default):
    for i, a in enumerate(sys.argv):
        if a == flag and i + 1 < len(sys.argv):
            return sys.argv[i + 1]
        if a.startswith(flag + "="):
            return a.split("=", 1)[1]
    return default


def self_launch():
    """``python bench.py --gpus N`` typed bare (no WORLD_SIZE / RANK in the environment) with N > 1: start N fresh rank processes -
    one per GPU, LOCAL_RANK = RANK = 0..N-1, rendezvous on 127.0.0.1 - before this process has made any GPU call (it never makes
    one), let rank 0's stdout (the one JSON line) through, wait, and exit with the worst child's status.  The first failing rank,
    or the time limit (--launch-timeout seconds, default 1500), ends the whole group.  Never an exec: children are subprocesses."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    try:
        n = int(_argv_value("--gpus", "1"))
    except ValueError:
        return                                   # argparse reports it
    if n <= 1 or "-h" in sys.argv or "--help" in sys.argv:
        return
    import signal
    import socket
    import subprocess

    limit = float(_argv_value("--launch-timeout", "1500"))
    if "--rehearse" not in sys.argv:
        # counted by a throw-away child, exactly as a rank will see them: THIS process never imports torch or touches HIP
        # (torch.cuda.device_count() falls back to hipGetDeviceCount, which initialises the runtime, when amdsmi does not import)
        try:
            ndev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                                      timeout=300).stdout.strip().splitlines()[-1])
        except (IndexError, ValueError, subprocess.TimeoutExpired):
            ndev = 0
        if ndev < n:
            raise SystemExit(f"bench.py: --gpus {n} means {n} ranks, one per GPU, but {ndev} GPU(s) are visible here "
                             "(--rehearse runs the multi-rank path with ranks sharing a GPU, labelled as a rehearsal)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ADAIN_SELF_LAUNCHED=str(os.getpid()))
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True,
                                      stdout=None if r == 0 else sys.stderr))

    def stop_all():
        for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            alive = [p for p in procs if p.poll() is None]
            for p in alive:
                try:
                    os.killpg(p.pid, sig)         # each child leads its own session: exactly the processes started here
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.time() + wait
            while time.time() < t_end and any(p.poll() is None for p in alive):
                time.sleep(0.05)

    def on_signal(signum, _frame):
        stop_all()
        os._exit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    t_end = time.time() + limit
    worst = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, c = bad[0]
            print(f"bench.py: rank {r} exited with status {c}: stopping the other ranks", file=sys.stderr, flush=True)
            stop_all()
            worst = c if c > 0 else 128 - c
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > t_end:
            print(f"bench.py: the {n} ranks did not finish within {limit:.0f} s: stopping them", file=sys.stderr, flush=True)
            stop_all()
            worst = 124
            break
        time.sleep(0.1)
    raise SystemExit(worst)


def watch_launcher():
    """A rank started by ``self_launch`` must not outlive it: if the launcher dies without stopping its ranks (SIGKILL), a daemon
    thread notices the re-parenting within a second and ends this process, so no orphan keeps a GPU busy."""
    pid = os.environ.get("ADAIN_SELF_LAUNCHED", "")
    if not pid.isdigit():
        return
    import threading

    def watch():
        while os.getppid() == int(pid):
            time.sleep(1.0)
        os._exit(70)

    threading.Thread(target=watch, name="adain-launcher-watch", daemon=True).start()


if __name__ == "__main__":
    self_launch()
    watch_launcher()

import numpy as np
import torch
import torch.distributed as dist

import applied_image_processing_amd.arch as arch
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.sharding as sh
from applied_image_processing_amd.telemetry import GpuTelemetry
import applied_image_processing_amd.synth as synth

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable by a float4 copy)
# The library runs the F(4,3) x F(2,3) kernels and nothing else.  `--diag-lib` (A/B runs only; the JSON line is labelled) loads the
# diagnostic build instead: the same sources with the environment tuning switches (ADAIN_W4_*, ADAIN_BIG_*) and timing-only variants.
if "--diag-lib" in sys.argv:
    sys.argv.remove("--diag-lib")
    rt.use_library(rt.DIAG_LIB_PATH)
if "--lib" in sys.argv:                  # same-box A/B against another build of the product library (the JSON line is labelled)
    _i = sys.argv.index("--lib")
    rt.use_library(os.path.abspath(sys.argv[_i + 1]))
    del sys.argv[_i:_i + 2]
_OTHER_LIB = os.path.basename(rt.LIB_PATH) if os.path.basename(rt.LIB_PATH) != "libadain_hip.so" else None
_DIAG_LIB = "diag" in os.path.basename(rt.LIB_PATH)
# multiplies the conv3x3 kernel executes on the matrix pipe per direct-convolution multiply
EXECUTED = 24.0 / 72.0
CONV_KERNEL = "conv3x3_wino4_kernel (Winograd F(4,3) x F(2,3))"
WORKLOADS = {
    2: "configs[1]: {h}x{w} AdaIN forward (style_transfer_simple), batch={b} per GPU, style {hs}x{ws} re-encoded every step, alpha={alpha}",
    3: "configs[2]: {h}x{w} depth-aware AdaIN (style_transfer, proximity-map blend), batch={b} per GPU, style {hs}x{ws} re-encoded every step",
    4: "configs[3]: one sub-batch of the video job per step, frames {h}x{w}, {b} frames per GPU per step, one {hs}x{ws} style (statistics cached per rank), uint8 out",
    5: "configs[4]: one sub-batch of the 3DGS guide-view job per step, views {h}x{w} with masks, {b} views per GPU per step, one {hs}x{ws} style, mask composite + uint8 out",
}
SIZES = {2: (1024, 1024), 3: (2048, 2048), 4: (1080, 1920), 5: (1200, 1600)}


def enc_conv3x3_flops(n, h, w):
    """Algorithmic flops (2 per MAC) of the 8 generic 3x3 conv launches of one encoder pass, in launch order."""
    out = []
    for L in arch.encoder_plan()[2:]:
        if L["src"] == "pool":
            h, w = (h + 1) // 2, (w + 1) // 2
        out.append(2 * n * h * w * L["cin"] * L["cout"] * 9)
    return out


def dec_conv3x3_flops(n, hc, wc):
    out = []
    for L in arch.decoder_plan()[:-1]:
        if L["src"] == "up":
            hc, wc = 2 * hc, 2 * wc
        out.append(2 * n * hc * wc * L["cin"] * L["cout"] * 9)
    return out


def make_events(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in evs:
        e.record()          # creates the underlying hipEvent_t; the C ABI re-records it
    return evs


class Timer:
    """HIP-event pairs around host-side calls on the current stream (the stream every kernel of the path is launched on)."""

    def __init__(self, on):
        self.on, self.spans = on, []

    def __call__(self, name, nbytes, fn, *a, **k):
        if not self.on:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.spans.append((name, nbytes, e0, e1))
        return r


def synth_frame(config, index, h, w):
    """Frame ``index`` of the synthetic job of ``config``: f32 [1,3,h,w] in [0,1); config 5 views carry ~30 %
    exact-zero background pixels (the reference's mask is ``gt_image_np > 0``, train.py:97)."""
    base = {2: 3, 3: 5, 4: 7, 5: 1000}[config]
    x = torch.from_numpy(synth.image(base + index, 1, h, w))
    if config == 5:
        bg = torch.from_numpy(synth.uniform01(2000 + index, h * w).reshape(1, 1, h, w) < 0.3)
        x = torch.where(bg, torch.zeros_like(x), x)
    return x


WEIGHT_SET = "kaiming"       # --weights: "kaiming" (zero-mean Kaiming-uniform, O(1) activations) or "trained-like" (synth.trained_like_state_dicts)


def synth_weights():
    vgg, dec = synth.state_dicts(WEIGHT_SET, 0)
    return synth.to_torch(vgg), synth.to_torch(dec)


JOB_FRAMES = {4: 512, 5: 300}        # BASELINE.json configs[3] / configs[4]


class FrameStore:
    """This rank's block [lo, hi) of a job's list of ``n`` decoded frames (uint8 [h,w,3]; SURVEY.md 8(d): video frames seed
    7 + i, guide views seed 1000 + i with ~30 % exact-zero background), generated on the GPU (``synth.frame_u8_torch``: the
    host generator's bits).  Resident in HBM it hands out views (``block``); with ``host=True`` the frames are pageable numpy
    arrays, as a decoder would leave them.  Touching a frame of another rank's block is an error."""

    def __init__(self, config, n, lo, hi, h, w, device, host=False):
        self.n, self.lo, self.hi, self.host = n, lo, hi, host
        base = {4: 7, 5: 1000}[config]
        dev = torch.empty((hi - lo, h, w, 3), dtype=torch.uint8, device=device)
        for k in range(lo, hi):
            dev[k - lo] = synth.frame_u8_torch(base + k, h, w, device, zero_fraction=0.3 if config == 5 else 0.0, zero_seed=2000 + k)
        torch.cuda.synchronize()
        self.dev = None if host else dev
        self.np = [dev[k].cpu().numpy() for k in range(hi - lo)] if host else None
        if not host:
            self.block = self._block             # only the HBM-resident store offers zero-copy blocks (jobs.FrameFeeder)

    def __len__(self):
        return self.n

    def _own(self, k):
        if not self.lo <= k < self.hi:
            raise IndexError(f"frame {k} belongs to another rank")
        return k - self.lo

    def __getitem__(self, k):
        return self.np[self._own(k)] if self.host else self.dev[self._own(k)]

    def _block(self, i, j):
        return self.dev[self._own(i):self._own(j - 1) + 1]


class MaskStore:
    """The masks of a guide-view job, ``view > 0`` as [3,h,w] (reference Style_3DGS/train.py:97), beside the frames: bool tensors
    in HBM or numpy arrays on the host."""

    def __init__(self, frames):
        self.f = frames
        if frames.host:
            self.m = [np.ascontiguousarray((a > 0).transpose(2, 0, 1)) for a in frames.np]
        else:
            self.m = (frames.dev > 0).permute(0, 3, 1, 2).contiguous()

    def __len__(self):
        return self.f.n

    def __getitem__(self, k):
        return self.m[self.f._own(k)]


class DepthStore:
    """Per-frame proximity maps of the video job's depth-aware variant ("a per-frame synthetic depth"): the
    smooth field of ``synth.smooth_depth`` with frame k's own 5 % noise (seed 6 + k), f32 [h,w] beside the frames (HBM or host)."""

    def __init__(self, frames, h, w, device):
        self.f = frames
        yy = np.arange(h, dtype=np.float64)[:, None] / max(h, 1)
        xx = np.arange(w, dtype=np.float64)[None, :] / max(w, 1)
        f = (np.sin(2 * math.pi * (1.0 * yy + 0.5 * xx)) + 0.7 * np.sin(2 * math.pi * (0.5 * yy - 1.5 * xx) + 1.0)
             + 0.5 * np.sin(2 * math.pi * (2.0 * yy + 1.0 * xx) + 2.0) + 0.3 * np.sin(2 * math.pi * (3.0 * xx) + 0.5))
        f = torch.from_numpy((f - f.min()) / (f.max() - f.min())).to(device)
        self.m = []
        for k in range(frames.lo, frames.hi):
            noise = synth.uniform01_torch(6 + k, h * w, device).reshape(h, w).double()
            d = ((0.95 * f + 0.05 * noise) * 1000.0).float()
            self.m.append(d.cpu().numpy() if frames.host else d)

    def __len__(self):
        return self.f.n

    def __getitem__(self, k):
        return self.m[self.f._own(k)]


class Step:
    """One pass of the hot path on device-resident inputs through the C ABI (see the module docstring)."""

    def __init__(self, device, config=2, first_frame=0, size=None, style_size=512, batch=1, alpha=0.5, engine=None, weights=None, depth=False):
        self.config, self.alpha, self.batch, self.device = config, alpha, batch, device
        self.h, self.w = SIZES[config]
        if size:
            self.h = self.w = size
        self.hs = self.ws = style_size
        self.first_frame = first_frame
        self.vgg_sd, self.dec_sd = weights if weights is not None else synth_weights()
        self.engine = engine if engine is not None else engine_mod.AdaINEngine(self.vgg_sd, self.dec_sd, device)      # packs the weights once
        self.enc, self.dec = self.engine.enc, self.engine.dec
        self.content = torch.cat([synth_frame(config, first_frame + i, self.h, self.w) for i in range(batch)]).to(device)
        self.style = torch.from_numpy(synth.image(4, 1, self.hs, self.ws)).to(device)
        self.hc, self.wc = rt.encoded_size(self.h, self.w)
        self.style_each_step = config in (2, 3)
        self.depth = None
        self.depth_offset = 0.15 if config == 3 else 0.30      # run_depth.py:33-40 defaults / the video caller's (video/utils.py:240)
        if config == 3 or depth:       # config 3, and the depth-aware variant of the video job (--depth)
            self.depth = [torch.from_numpy(synth.smooth_depth(6 + first_frame + i, self.h, self.w)).to(device) for i in range(batch)]
        if config == 5:
            self.mask = (self.content > 0).float()
        if not self.style_each_step:
            self.s_mean, self.s_std = self.engine.set_style(self.style).style_stats()
            self.style_cache = {0: (self.s_mean, self.s_std)}      # one style for the whole job: encoded once per rank
        self.spans = []

    def flops_per_step(self):
        f = arch.conv_flops_encoder(self.h, self.w) * self.batch + arch.conv_flops_decoder(self.hc, self.wc) * self.batch
        if self.style_each_step:
            f += arch.conv_flops_encoder(self.hs, self.ws)
        return f

    def conv3x3_algorithmic_bytes(self):
        """input + output + weights of every generic 3x3 conv launch of a step (what a launch must move at least)."""
        def enc(n, h, w):
            out = []
            for L in arch.encoder_plan()[2:]:
                if L["src"] == "pool":
                    h, w = (h + 1) // 2, (w + 1) // 2
                pooled = L["idx"] in (5, 12, 25)       # conv1_2, conv2_2, conv3_4 write only the pooled tensor
                oh, ow = ((h + 1) // 2, (w + 1) // 2) if pooled else (h, w)
                out.append(4 * (n * h * w * L["cin"] + n * oh * ow * L["cout"] + 9 * L["cin"] * L["cout"]))
            return out

        def dec(n, h, w):
            out = []
            for L in arch.decoder_plan()[:-1]:
                ih, iw = h, w
                if L["src"] == "up":
                    h, w = 2 * h, 2 * w
                out.append(4 * (n * ih * iw * L["cin"] + n * h * w * L["cout"] + 9 * L["cin"] * L["cout"]))
            return out

        b = enc(self.batch, self.h, self.w)
        if self.style_each_step:      # content and style share one launch per encoder layer (adain_encode_multi)
            b = [x + y - 4 * 9 * L["cin"] * L["cout"] for x, y, L in zip(b, enc(1, self.hs, self.ws), arch.encoder_plan()[2:])]
        return b + dec(self.batch, self.hc, self.wc)

    def conv3x3_flops(self):
        f = enc_conv3x3_flops(self.batch, self.h, self.w)
        if self.style_each_step:
            f = [x + y for x, y in zip(f, enc_conv3x3_flops(1, self.hs, self.ws))]
        return f + dec_conv3x3_flops(self.batch, self.hc, self.wc)

    def run(self, timed=False, to_u8=False, u8_out=None):
        """Returns the float result (and the conv event pairs when ``timed``); ``self.u8`` holds the uint8 frames when the
        config quantises (4, 5) or ``to_u8`` asks for it (written into ``u8_out`` [b,H,W,3] when given)."""
        ev = []
        self.edge_ev = []
        T = Timer(timed)
        n, hc, wc = self.batch, self.hc, self.wc
        px = n * self.h * self.w
        feat_bytes = n * hc * wc * 512 * 4
        ev_c = make_events(11) if timed else None
        if self.style_each_step:
            # content and style through the encoder in one pass (what style_transfer / style_transfer_simple do): one launch per layer
            cf, sf = rt.encode_multi([self.content, self.style], self.enc, ev_c)
            s_mean, s_std = rt.mean_std(sf, True)
            first_px = px + self.hs * self.ws
        else:
            cf = rt.encode(self.content, self.enc, ev_c)
            s_mean, s_std = self.s_mean, self.s_std
            first_px = px
        if timed:
            ev += [(ev_c[i + 1], ev_c[i + 2]) for i in range(8)]
            self.edge_ev.append(("conv_first_kernel (NCHW image -> 64-ch NHWC)", first_px * (12 + 256), ev_c[0], ev_c[1]))
        c_mean, c_std = T("mean_std_nhwc_partial + finalize (content relu4_1)", feat_bytes, rt.mean_std, cf, True)
        if self.depth is not None:
            p = torch.cat([T("bicubic_minmax + strength_sum + strength_apply (P map)", self.h * self.w * 4 + hc * wc * 12,
                             rt.strength_map, d, hc, wc, self.depth_offset, 20) for d in self.depth])
            g = T("adain_blend_kernel (P-map blend)", 2 * feat_bytes, rt.blend_pmap, cf, True, c_mean, c_std, s_mean, s_std, p)
        else:
            g = T("adain_blend_kernel (alpha blend)", 2 * feat_bytes, rt.blend_alpha, cf, True, c_mean, c_std, s_mean, s_std, self.alpha)
        ev_d = make_events(10) if timed else None
        out = rt.decode(g, self.dec, ev_d)
        if timed:
            ev += [(ev_d[i], ev_d[i + 1]) for i in range(8)]
            self.edge_ev.append(("conv_last_kernel (64-ch NHWC -> NCHW image)", px * (256 + 12), ev_d[8], ev_d[9]))
        if self.config == 5:
            # the decoder's output and the mask already have the view's size (1200 x 1600: multiples of 8; mask = view > 0): the two
            # F.interpolate calls of test.py:222-236 are identities and the engine skips them (engine.AdaINEngine.composite)
            out = T("mask_composite_kernel", 4 * px * 12, rt.mask_composite, self.content, out, self.mask)
        if self.config in (4, 5) or to_u8:
            self.u8 = T("quantize_u8_kernel", px * 15, rt.quantize_u8, out, u8_out)
        self.spans = T.spans
        return (out, ev) if timed else out


def load_pmc_traffic(workload_key):
    """HBM bytes per conv3x3 launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/summarize_rocprof.py --traffic from two separate --pmc runs of this bench command); (None, None) when no profile
    of this workload is committed.  A constant taken from a profile of the same command, not a live measurement."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(workload_key, {})
        return d.get("hbm_bytes_per_conv3x3_launch"), d.get("source")
    except Exception:
        return None, None


def measure_roofline(step, reps):
    flops = step.conv3x3_flops()
    total_ms = 0.0
    per_layer = [0.0] * len(flops)
    sec = {}
    # back to the chip's sustained clock after the host work in between: it needs about 8 steps of load to get there after idling
    # (profiles/r04_step_ramp_after_idle.json), and a fraction of the PEAK rate is only meaningful at the clock the peak is quoted for.
    # (This is the instrumented leg: `value` was measured before it, under the contract's own W and K.)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.12:
        for _ in range(4):
            step.run()
        torch.cuda.synchronize()
    step.run(timed=True)
    torch.cuda.synchronize()
    for _ in range(reps):
        _, ev = step.run(timed=True)
        torch.cuda.synchronize()
        d = [a.elapsed_time(b) for a, b in ev]
        for name, nbytes, a, b in step.edge_ev + step.spans:
            s = sec.setdefault(name, [nbytes, 0.0, 0])
            s[1] += a.elapsed_time(b)
            s[2] += 1
        per_layer = [x + y for x, y in zip(per_layer, d)]
        total_ms += sum(d)
    launches = len(flops) * reps
    avg_ms = total_ms / launches
    algorithmic = sum(flops) * reps / (total_ms * 1e-3) / 1e12
    executed = algorithmic * EXECUTED
    layers = [{"gflop": f / 1e9, "ms": t / reps, "tflops": f / (t / reps * 1e-3) / 1e12} for f, t in zip(flops, per_layer)]
    traffic, traffic_src = load_pmc_traffic(f"config{step.config}_batch{step.batch}")
    if (step.h, step.w) != SIZES[step.config] or (step.hs, step.ws) != (512, 512):
        traffic, traffic_src = None, None          # the committed counter passes are of the config's own sizes
    roof = {
        "bound": "mfma", "achieved": round(executed, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "kernel": CONV_KERNEL + f" ({len(flops)} launches/step)",
        "avg_launch_ms": round(avg_ms, 4),
        # `achieved` counts the multiplies the kernel executes on the matrix pipe: the direct 3x3 convolution's
        # divided by `multiply_reduction`; `algorithmic_tflops` = direct-convolution flops / time (may exceed the peak)
        "multiply_reduction": round(1.0 / EXECUTED, 3),
        "executed_gflop_per_launch_avg": round(sum(flops) * EXECUTED / len(flops) / 1e9, 3),
        "algorithmic_gflop_per_launch_avg": round(sum(flops) / len(flops) / 1e9, 3),
        "algorithmic_tflops": round(algorithmic, 2),
        "algorithmic_bytes_per_launch_avg": sum(step.conv3x3_algorithmic_bytes()) / len(flops),
    }
    if step.batch > 1:
        roof["note"] = ("batch > 1: the instrumented passes (per-layer events) run every layer once over the whole batch; the timed steps run a "
                        "batch's big layers frame by frame (csrc/api.hip, BIG_LAYER_ROUNDS), which is faster for >= 2 Mpixel frames")
    secondary = []
    for name, (nbytes, ms, cnt) in sec.items():
        us = ms / cnt * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9
        secondary.append({"kernel": name, "bound": "hbm", "algorithmic_bytes": nbytes, "avg_us": round(us, 2),
                          "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
    return roof, layers, secondary


def measure_pixel_kernels(device, reps=10):
    """The frame-sized pixel kernels of the video / guide-view jobs on a batch of 8 1080p frames,
    each launched over rotating buffer sets of more than 512 MiB in total, so that nothing is served from the 256 MiB
    Infinity Cache: algorithmic bytes / HIP-event time per launch."""
    n, h, w, sets = 8, 1080, 1920, 3
    px = n * h * w
    gen = torch.Generator(device="cpu").manual_seed(11)
    f32 = [torch.rand((n, 3, h, w), generator=gen).to(device) for _ in range(sets)]
    f32b = [x.flip(0) for x in f32]
    msk = [(x > 0.3).float() for x in f32]
    u8 = [rt.quantize_u8(x) for x in f32]
    flow = [(torch.rand((2, h, w), generator=gen) * 6 - 3).to(device) for _ in range(sets)]
    cases = [
        ("quantize_u8_kernel", px * 15, lambda i: rt.quantize_u8(f32[i])),
        ("mask_composite_kernel", px * 48, lambda i: rt.mask_composite(f32[i], f32b[i], msk[i])),
        ("resize_bilinear_kernel (same size)", px * 24, lambda i: rt.resize_bilinear(f32[i], (h, w))),
        ("resize_nearest_kernel (same size)", px * 24, lambda i: rt.resize_nearest(msk[i], (h, w))),
        ("resize_area2x2_rgb4_kernel (1080p -> 540x960)", px * 3 * 5 // 4, lambda i: rt.resize_area_u8(u8[i], (w // 2, h // 2))),
        ("resize_area_tab_u8_kernel (1080p -> 720x1280, fractional)", px * 3 * 13 // 9, lambda i: rt.resize_area_u8(u8[i], (1280, 720))),
        ("warp_blend_u8_kernel (one 1080p frame)", h * w * (9 + 8), lambda i: rt.warp_blend_u8(u8[i][0], u8[i][1], flow[i], 0.7)),
    ]
    out = []
    for name, nbytes, fn in cases:
        for i in range(sets):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            fn(r % sets)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9
        out.append({"kernel": name, "bound": "hbm", "algorithmic_bytes": nbytes, "avg_us": round(us, 2), "achieved": round(gbs, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "batch": "8 x 1080x1920"})
    return out


def cpu_baseline(step, gpu_out, job_frames=None, job_depth=None):
    """The oracle on this node's host cores, same workload, bounded sample (about 10-30 s): configs 2 / 3 whole forwards of
    the step's first frame; configs 4 / 5 two frames of the job (per-frame cost is constant)."""
    from oracle import adain_oracle as O

    # threads actually used: the GPU box grants a 1-GPU job a CPU share of 16 cores (more threads than that
    # oversubscribe and run several times slower); ADAIN_CPU_THREADS overrides
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("ADAIN_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    cfg, h, w = step.config, step.h, step.w
    vgg_sd, dec_sd = step.vgg_sd, step.dec_sd
    s = step.style.cpu()
    if job_frames is not None:     # --job: the job's first decoded frames (uint8 HWC) through ToTensor, as the reference loads them
        frames = [f.cpu().permute(2, 0, 1).float().div(255).unsqueeze(0) for f in job_frames]
    else:
        frames = [step.content[i:i + 1].cpu() for i in range(min(step.batch, 2))]
        if cfg in (4, 5) and len(frames) < 2:
            frames.append(synth_frame(cfg, step.first_frame + 1, h, w))

    depth = None
    if job_depth is not None:
        depth = [torch.as_tensor(d).cpu() for d in job_depth]
    elif step.depth is not None:
        depth = [d.cpu() for d in step.depth[:2]]
        if len(depth) < len(frames):
            depth.append(torch.from_numpy(synth.smooth_depth(6 + step.first_frame + 1, h, w)))

    def forward(i):
        c = frames[i]
        if cfg == 3:
            return O.style_transfer(vgg_sd, dec_sd, c, s, depth[i], 1.0, step.depth_offset, 20)
        if depth is not None:
            out = O.style_transfer(vgg_sd, dec_sd, c, s, depth[i], 1.0, step.depth_offset, 20)
        else:
            out = O.style_transfer_simple(vgg_sd, dec_sd, c, s, step.alpha)
        if cfg == 5:
            out = O.mask_composite(c, out, (c[0] > 0))
        return O.quantize_u8(out) if cfg in (4, 5) else out

    times = []
    with torch.no_grad():
        ref = forward(0)                       # warm-up, also the PSNR reference for frame 0
        t_end = time.time() + 20.0
        k = 0
        while len(times) < 2 or (time.time() < t_end and len(times) < 5):
            t0 = time.perf_counter()
            forward(k % len(frames))
            times.append(time.perf_counter() - t0)
            k += 1
    best = min(times)
    if cfg in (4, 5):
        got = gpu_out[0].cpu()                 # uint8 HWC
        diff = (got.int() - ref[0].int()).abs()
        rel = float(diff.float().norm() / ref[0].float().norm())
        mse = float((diff.float() / 255.0).pow(2).mean())
        psnr = float("inf") if mse == 0 else 10 * torch.log10(torch.tensor(1.0 / mse)).item()
    else:
        got = gpu_out[:1].cpu()
        rel = float((got - ref).norm() / ref.norm())
        psnr = float(O.psnr(got.clamp(0, 1), ref.clamp(0, 1)).min())
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    what = {2: "full style_transfer_simple forwards", 3: "full depth-aware style_transfer forwards",
            4: ("depth-aware " if depth is not None else "") + "video-job frames (forward + uint8), per-frame rate", 5: "guide views (forward + mask composite + uint8), per-view rate"}[cfg]
    return {
        "value": round(h * w / 1e6 / best, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"{len(times)} {what} of the same workload ({h}x{w} content, {step.hs}x{step.ws} style"
                  f"{' encoded once' if cfg in (4, 5) else ''}), best of {len(times)}; median {statistics.median(times):.2f} s; "
                  f"torch {torch.__version__} CPU, {model}",
    }, psnr, rel


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 3 whole jobs with --job)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5; 1 job with --job)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] index + 1 (default 2 = configs[1])")
    ap.add_argument("--size", type=int, default=0, help="override content H = W")
    ap.add_argument("--style-size", type=int, default=512, help="style H = W (512 = the API default style_size; SURVEY 8(d) also names the 1024 variant of config 2)")
    ap.add_argument("--alpha", type=float, default=0.5, help="configs 2 / 4 / 5: content-style trade-off of the alpha blend (default 0.5; SURVEY 8(d) also names 1.0)")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 1); with --job: frames per sub-batch (default: chosen by the driver from the frame size, jobs.auto_sub_batch: 4 at 1080p and 1200 x 1600 - the C schedules run a batch's big layers frame by frame -, about three megapixels for smaller frames: 256 x 456 frames peak at 16 - 32)")
    ap.add_argument("--depth", action="store_true", help="config 4: the depth-aware variant (a synthetic proximity map per frame, use_depth=True of the reference's video caller: offset 0.30, prominence 20)")
    ap.add_argument("--job", action="store_true", help="configs 4 / 5: run the BASELINE job (512 frames / 300 views) strong-scaled over the ranks")
    ap.add_argument("--frames", type=int, default=0, help="--job: frames of the whole job (default 512 / 300)")
    ap.add_argument("--host-frames", action="store_true",
                    help="--job: also time the job with its frames in host memory (pinned staging + copy stream) and the result copied back")
    ap.add_argument("--gather-chunks", type=int, default=1, help="--job: pieces the one gather is issued in (overlapping the remaining compute)")
    ap.add_argument("--gather", choices=list(jobs.GATHER_MODES), default="end",
                    help="more than one rank, per-step mode: 'end' (default) = ONE gather of every rank's frames of the K timed steps at the end "
                         "of the timed region (no transport kernel beside the compute kernels); 'overlap' = one asynchronous gather per step")
    ap.add_argument("--sustain", type=float, default=6.0, help="single GPU, per-step mode: also run the same loop for at least this many seconds "
                    "after the timed region and report it as `sustained` (0 = skip)")
    ap.add_argument("--per-call", choices=["video", "guide"], default=None,
                    help="time the reference's unchanged caller loops (one adain_inference call per frame / view, files included) instead of the "
                         "BASELINE step; --steps = calls (default 20)")
    ap.add_argument("--schedule", choices=["batch", "latency"], default="batch",
                    help="--per-call: `latency` runs adain_inference under ADAIN_SCHEDULE_LATENCY (cin split of the layers one small frame under-fills; "
                         "AdaIN.test.set_latency_schedule)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="bare `--gpus N` launch: seconds after which the rank processes are stopped")
    ap.add_argument("--n1-value", type=float, default=0.0, help="--job: the 1-GPU value of the same job, to report efficiency_vs_n1")
    ap.add_argument("--weights", choices=["kaiming", "trained-like"], default="kaiming",
                    help="the seeded weight set: zero-mean Kaiming (default; every round-1..4 number) or the trained-like statistics of "
                         "synth.trained_like_state_dicts (the regime of the checkpoint the reference loads; same kernels, same flops)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1080p pixel-kernel bandwidth table")
    ap.add_argument("--pcie", action="store_true", help="also report the rate with the frame crossing PCIe both ways (never `value`)")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--rehearse", action="store_true",
                    help="single-GPU rehearsal of the multi-rank path: ranks may share a GPU and the gather may run over gloo (labelled in the JSON)")
    args = ap.parse_args()
    if args.job and args.config not in (4, 5):
        ap.error("--job runs the video job (--config 4) or the guide-view job (--config 5)")
    if args.depth and args.config != 4:
        ap.error("--depth is the depth-aware variant of the video job (--config 4); config 3 is depth-aware by itself")
    if args.steps is None:
        args.steps = 3 if args.job else 20
    if args.warmup is None:
        args.warmup = 1 if args.job else 5
    if args.batch is None:
        args.batch = 0 if args.job else 1          # --job: 0 = the driver's automatic sub-batch (4 frames of 1080p / 1200 x 1600, about three megapixels of smaller frames)
    return args


class Ctx:
    """Process / device / process-group setup shared by the per-step and the job mode."""

    def __init__(self, args):
        self.world = world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != world:
            if world == 1 and args.gpus > 1:      # (unreachable from the command line: self_launch() starts the ranks)
                raise SystemExit("bench.py --gpus N runs N rank processes: start it as a script, or under torch.distributed.run")
            args.gpus = world
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the AdaIN path has no CPU fallback)")
        ndev = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if local_world > ndev and not args.rehearse:
            raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} GPU(s) visible: one process per GPU is the "
                             "contract (RCCL rejects two ranks on one device); use --rehearse for a single-GPU rehearsal")
        self.shared_gpu = local_world > ndev
        self.rehearse = bool(args.rehearse)
        torch.cuda.set_device(local_rank % ndev)
        self.device = torch.device("cuda", local_rank % ndev)
        self.use_dist = world > 1 or "RANK" in os.environ      # torch.distributed.run sets RANK even for one process
        self.transport = None
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # gloo carries the host-side rendezvous (barriers, the MAX of the step time); RCCL ("nccl") carries device
            # tensors, i.e. the gather of the finished frames over xGMI.  With ranks sharing a GPU (--rehearse) RCCL cannot
            # start, so the rehearsal gathers host copies over gloo: chosen here, up front, identically on every rank.
            dist.init_process_group("gloo" if self.shared_gpu else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)
            self.transport = "gloo" if self.shared_gpu else sh.device_transport(torch.empty(0, dtype=torch.uint8, device=self.device))
            if self.transport != "rccl" and not args.rehearse:
                raise SystemExit(f"bench.py: the device gather would run over {self.transport!r}, not RCCL: refusing to report a multi-GPU number")
        self.ranks = self.describe_ranks(local_rank % ndev)

    def describe_ranks(self, dev_index):
        """What the JSON line needs to show that the device transport really saw ``world`` ranks on ``world`` devices: every rank's
        device (index, uuid, name), pid and host, collected with all_gather_object, the RCCL version, and the result of ONE
        all_reduce(SUM) of a device-resident 1 per rank over the device transport (= world if every rank took part)."""
        props = torch.cuda.get_device_properties(dev_index)
        mine = {"rank": self.rank, "device": dev_index, "uuid": str(getattr(props, "uuid", "")), "name": props.name,
                "gcn_arch": getattr(props, "gcnArchName", ""), "pid": os.getpid(), "host": os.uname().nodename}
        if not self.use_dist:
            return {"world": 1, "devices": [mine], "transport": None, "launcher": "single process"}
        census = sh.rank_census(mine, None if self.transport != "rccl" else self.device)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if self.transport == "rccl" else None
        except Exception as e:                        # the proof above does not depend on it
            ver = f"unavailable ({type(e).__name__})"
        census.update(transport=self.transport, rccl=ver, backend=str(dist.get_backend_config()),
                      launcher="bench.py self-launch" if os.environ.get("ADAIN_SELF_LAUNCHED") else "torch.distributed.run / external")
        # a line whose transport did not see every rank, or whose ranks share a device, is not a multi-GPU measurement: every rank
        # holds the same census, so every rank stops here, non-zero, before anything is timed (a rehearsal is labelled instead)
        problems = sh.census_problems(census, shared_devices_allowed=self.rehearse)
        if problems:
            if self.rank == 0:
                print("bench.py: refusing to report a multi-GPU number: " + "; ".join(problems), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            raise SystemExit(3)
        return census

    def barrier(self):
        torch.cuda.synchronize()                      # this rank's GPU work is done ...
        if self.use_dist:
            dist.all_reduce(torch.zeros(1))           # ... and so is everybody else's (host rendezvous, gloo)
        torch.cuda.synchronize()

    def warm_transport(self, n_job, batch):
        """Connection setup of the device transport (RCCL builds its communicator on first use) stays out of every timed region."""
        if self.use_dist:
            probe = torch.zeros((batch, 8, 8, 3), dtype=torch.uint8, device=self.device)
            sh.gather_frames(probe, n_job, dst=0, counts=[batch] * self.world)
            self.barrier()

    # How a multi-rank run ENDS.  After the per-rank table rank 0 still has host-only legs to run (the instrumented roofline frame, the
    # pixel-kernel table, the CPU baseline: 15 s and more) that need nobody else.  The other ranks must neither sit in a collective
    # meanwhile (its timeout is not ours to know: round-5 advisor finding) nor tear their communicators down while rank 0 still holds
    # its own (RCCL with N > 1 has never run here: no untested shutdown orders).  So they release their GPU memory and wait on a KEY of
    # the rendezvous store with an explicit, generous timeout; rank 0 sets it when its line is out; then every rank destroys the group.
    DONE_KEY = "adain_bench_rank0_done"

    def others_wait_for_rank0(self, timeout_s):
        """Ranks != 0, once their part of the measurement is over: free the device memory, then block on the store key (not a collective)."""
        import datetime
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        dist.distributed_c10d._get_default_store().wait([self.DONE_KEY], datetime.timedelta(seconds=max(60.0, float(timeout_s))))

    def finish(self, rank0_done=False):
        if self.use_dist:
            if rank0_done and self.rank == 0:
                dist.distributed_c10d._get_default_store().set(self.DONE_KEY, b"1")
            dist.destroy_process_group()


def base_result(args, ctx, value, ms, workload, parallelism, scaling):
    return {
        **({"diagnostic_library": True} if _DIAG_LIB else {}), **({"library": _OTHER_LIB} if _OTHER_LIB else {}),
        "metric": "stylized Mpixels/sec, AdaIN forward (encode content + encode style + AdaIN + decode)",
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload + ", fp32, seeded synthetic weights (reference architecture"
                               + (")" if WEIGHT_SET == "kaiming" else "; TRAINED-LIKE statistics: Caffe-style conv0, non-zero-mean / zero-sum filters, "
                                                                      "channels normalised to a post-ReLU mean near 1)"), "parallelism": parallelism},
    }


def main_job(args, ctx):
    """``--job``: the BASELINE job of config 4 / 5 over all ranks (see the module docstring)."""
    world, rank, device = ctx.world, ctx.rank, ctx.device
    cfg = args.config
    h, w = SIZES[cfg]
    hs = ws = args.style_size
    n_total = args.frames or JOB_FRAMES[cfg]
    sub = args.batch or jobs.auto_sub_batch(h, w)          # frames per sub-batch (what the driver picks itself when given None)
    lo, hi = sh.shard_range(n_total, world, rank)
    weights = synth_weights()
    engine = engine_mod.AdaINEngine(weights[0], weights[1], device)
    style = torch.from_numpy(synth.image(4, 1, hs, ws)).to(device)
    style_cache = {}
    via_rccl = ctx.use_dist and ctx.transport == "rccl"       # (--rehearse: ranks share a GPU, the gather moves host copies over gloo)

    def make_job(frames, masks, host_result=None, depth_maps=None):
        def job():
            res, info = jobs.stylize_frames_sharded(engine, frames, style, alpha=args.alpha, masks=masks, depth_maps=depth_maps, depth_offset=0.30,
                                                    depth_prominence=20, sub_batch=args.batch or None, gather=ctx.use_dist,
                                                    require_transport="rccl" if via_rccl and world > 1 else None, style_cache=style_cache,
                                                    out_hw=(h, w), gather_chunks=args.gather_chunks,
                                                    host_out=host_result)      # the finished frames leave the device behind the kernels
            return res, info
        return job

    frames = FrameStore(cfg, n_total, lo, hi, h, w, device, host=False)
    masks = MaskStore(frames) if cfg == 5 else None
    depths = DepthStore(frames, h, w, device) if args.depth else None
    ctx.warm_transport(world, 1)
    telemetry = GpuTelemetry(device.index).start()
    dt, res, info = jobs.run_timed_jobs(make_job(frames, masks, depth_maps=depths), args.steps, args.warmup, barrier=ctx.barrier)
    telemetry.window("timed_region", info["t0"], info["t1"])
    if rank == 0:
        assert res is not None and res.shape == (n_total, h, w, 3), (None if res is None else res.shape)
    first_u8 = res[:2].clone() if rank == 0 else None
    per_rank = [None] * world
    mine = {k: round(float(info[k]), 4) for k in ("compute_s", "gather_s", "enqueue_s", "fetch_s", "host_cpu_s", "process_cpu_s")}
    mine["frames"] = hi - lo
    if hi > lo:
        # CPU the launching thread / the whole process spend per frame (eight ranks share one host), and C-ABI calls per frame
        mine["launch_thread_cpu_ms_per_frame"] = round(info["host_cpu_s"] * 1e3 / (hi - lo), 4)
        mine["process_cpu_ms_per_frame"] = round(info["process_cpu_s"] * 1e3 / (hi - lo), 4)
        mine["abi_calls_per_frame"] = round(info["abi_calls"] / (hi - lo), 2)
    mine["gpu"] = telemetry.stop()          # this rank's shader clock and package power over the timed jobs (sysfs, side thread)
    if ctx.use_dist:
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]

    pcie = None
    if args.host_frames:
        del res
        host_frames = FrameStore(cfg, n_total, lo, hi, h, w, device, host=True)
        host_masks = MaskStore(host_frames) if cfg == 5 else None
        host_depths = DepthStore(host_frames, h, w, device) if args.depth else None
        host_out = torch.empty((n_total, h, w, 3), dtype=torch.uint8).pin_memory() if rank == 0 else None
        pdt, pres, pinfo = jobs.run_timed_jobs(make_job(host_frames, host_masks, host_out, host_depths), args.steps, args.warmup, barrier=ctx.barrier)
        if rank == 0:
            assert torch.equal(host_out[:2], first_u8.cpu()), "host-resident job differs from the HBM-resident one"
        pcie = {"value": round(n_total * h * w / 1e6 / (pdt / args.steps), 3), "unit": "Mpixels/s", "ms_per_step": round(pdt / args.steps * 1e3, 3),
                "what": "the same job with its frames (and masks) in pageable host memory: pinned staging + uint8 upload on a copy stream "
                        "behind the kernels, ToTensor on the device, the gathered uint8 result copied back to pinned host memory",
                "h2d_bytes_rank0": int(pinfo["h2d_bytes"]), "d2h_bytes_rank0": int(pinfo.get("d2h_bytes", 0)),
                "rank0": {k: round(float(pinfo[k]), 4) for k in ("compute_s", "gather_s", "enqueue_s", "fetch_s")}, "feeder_rank0": pinfo.get("feeder"),
                "bit_identical_to_resident": True}

    if rank != 0:          # rank 0's host-only legs (roofline frame, pixel-kernel table, CPU baseline) need nobody else: see Ctx.finish
        ctx.others_wait_for_rank0(args.launch_timeout)
        ctx.finish()
        return
    if rank == 0:
        sec_per_job = dt / args.steps
        value = n_total * h * w / 1e6 / sec_per_job
        what = {4: f"configs[3]: video job, {n_total} frames {h}x{w}" + (", depth-aware (a proximity map per frame, offset 0.30, prominence 20)" if args.depth else ""), 5: f"configs[4]: 3DGS guide-view job, {n_total} views {h}x{w} with masks"}[cfg]
        shard = sh.shard_counts(n_total, world)
        workload = (f"{what}, one {hs}x{ws} style (statistics cached per rank), sub-batches of {sub}{' (automatic)' if not args.batch else ''}, decoded uint8 frames resident in HBM, "
                    f"{'mask composite + ' if cfg == 5 else ''}uint8 out; a step = the whole job")
        parallelism = (f"frame sharding x{world}: contiguous blocks {shard if world > 1 else ''} per rank, replicated weights and style statistics, "
                       f"one status word + ONE gather of the uint8 frames to rank 0 per job" if ctx.use_dist else "single GPU, no collective")
        result = base_result(args, ctx, value, sec_per_job * 1e3, workload, parallelism, "strong")
        # the roofline of a job line is taken on ONE frame: the C schedules run a sub-batch's big layers frame by frame (and the
        # per-layer events of the instrumented run would switch that schedule off for a batch)
        step = Step(device, config=cfg, first_frame=0, style_size=args.style_size, batch=1, engine=engine, weights=weights, depth=args.depth, alpha=args.alpha)
        roof, layers, secondary = measure_roofline(step, 5)
        roof["measured_on"] = "one frame of the job (batch 1), HIP events per conv launch"
        result["roofline"] = roof
        result["secondary"] = secondary
        result["step_tflops"] = round((step.flops_per_step() * n_total) / sec_per_job / 1e12, 2)
        result["job"] = {"driver": "jobs.stylize_frames_sharded", "frames": n_total, "frames_per_rank": shard, "sub_batch": sub,
                         "gathers_per_job": info["gathers"], "gather_chunks": args.gather_chunks, "transport": ctx.transport,
                         "per_rank": per_rank, "ms_per_frame": round(sec_per_job * 1e3 / max(shard), 4)}
        result["ranks"] = ctx.ranks
        if args.n1_value > 0:
            result["job"]["efficiency_vs_n1"] = round(value / (args.n1_value * world), 4)
        if args.rehearse:
            result["rehearsal"] = {"ranks_share_a_gpu": ctx.shared_gpu, "note": "not a multi-GPU measurement"}
        if pcie is not None:
            result["pcie_inclusive"] = pcie
        if not args.no_secondary:              # rank 0's GPU, after the timed region (the other ranks wait on the store key)
            result["secondary"] += measure_pixel_kernels(device)
        if not args.no_cpu:                    # rank 0's host cores, after the timed region, at every N (north_star: "in the same run")
            step.run()
            torch.cuda.synchronize()
            cb, psnr, rel = cpu_baseline(step, first_u8, job_frames=[frames[k] for k in range(min(2, n_total))],
                                          job_depth=[depths[k] for k in range(min(2, n_total))] if depths is not None else None)
            if world > 1:
                cb["sample"] += f"; measured on rank 0 after the timed region (the other {world - 1} rank(s) have freed their device memory and wait on a key of the rendezvous store, in no collective)"
            result["cpu_baseline"] = cb
            result["psnr_db_vs_cpu"] = round(psnr, 2) if psnr != float("inf") else "inf"
            result["rel_l2_vs_cpu"] = float(f"{rel:.3e}")
        if args.layers and layers:
            for i, L in enumerate(layers):
                print(f"layer {i:2d}: {L['gflop']:8.2f} GF  {L['ms']:8.4f} ms  {L['tflops']:7.2f} TF/s", file=sys.stderr)
        emit(result)
    ctx.finish(rank0_done=True)


def main_per_call(args, ctx):
    """``--per-call video|guide``: what the reference's UNCHANGED callers get from one ``adain_inference`` call - the loop of
    video/utils.py:341-350 (frame files in, ``content_size=256``, ``use_depth=True`` with a proximity map, JPEG out) or of
    Style_3DGS/train.py:86-115 (PIL views in, ``content_size=512``, mask ``view > 0``, JPEG out) - with a per-stage breakdown, next
    to the call-by-call path that re-encodes the style every time (as the reference does) and to the oracle on the host cores.
    Not the BASELINE metric: a latency / calls-per-second line of its own (file I/O and PIL are inside, as they are for the caller)."""
    import tempfile

    from PIL import Image

    from applied_image_processing_amd.AdaIN import test as T

    mode, n = args.per_call, max(4, args.steps)
    T.set_latency_schedule(args.schedule == "latency")
    h, w = ((270, 480) if mode == "video" else (800, 800)) if not args.size else (args.size, args.size)     # the callers' own shapes
    csize = 256 if mode == "video" else 512
    root = tempfile.mkdtemp(prefix="adain_per_call_")
    vgg_sd, dec_sd = synth_weights()
    torch.save(vgg_sd, os.path.join(root, "vgg.pth"))
    torch.save(dec_sd, os.path.join(root, "dec.pth"))
    ck = dict(vgg_str=os.path.join(root, "vgg.pth"), decoder_str=os.path.join(root, "dec.pth"))
    frames = [(synth.image(7 + k, 1, h, w)[0].transpose(1, 2, 0) * 255).astype(np.uint8) for k in range(n)]
    style_arr = (synth.image(4, 1, 700, 933)[0].transpose(1, 2, 0) * 255).astype(np.uint8)      # the sample style's size
    style_path = os.path.join(root, "style.jpg")
    Image.fromarray(style_arr).save(style_path, quality=95)
    os.makedirs(os.path.join(root, "frames"))
    for k, a in enumerate(frames):
        if mode == "guide":
            a[synth.uniform01(2000 + k, h * w).reshape(h, w) < 0.3] = 0
        Image.fromarray(a).save(os.path.join(root, "frames", f"frame_{k:04d}.jpg"), quality=95)
    depth = [torch.from_numpy(synth.smooth_depth(6 + k, h, w)) for k in range(n)] if mode == "video" else None
    style_obj = Image.open(style_path)
    style_obj.load()

    def call(k, out):
        if mode == "video":
            return T.adain_inference(os.path.join(root, "frames", f"frame_{k:04d}.jpg"), style_path, content_size=256, output=os.path.join(root, out),
                                     file_name=f"frame_{k:04d}", depth_offset=0.30, depth_prominence=20, use_depth=True, depth_map=depth[k], **ck)
        return T.adain_inference(content_img=Image.fromarray(frames[k]), style_img=style_obj, content_size=512, style_size=512,
                                 content_mask=frames[k].transpose(2, 0, 1) > 0, output=os.path.join(root, out), file_name=f"view_{k:04d}", **ck)

    def timed_loop(out, cache):
        import contextlib
        import io

        T.clear_style_cache()
        T.set_style_cache(cache)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            for k in range(min(args.warmup, n)):
                call(k, out)
            torch.cuda.synchronize()
            T._stage_timer.reset()
            T._stage_timer.on = cache
            c0, p0, t0 = time.thread_time(), time.process_time(), time.perf_counter()
            paths = [call(k, out) for k in range(n)]
            torch.cuda.synchronize()
            dt, cpu, pcpu = time.perf_counter() - t0, time.thread_time() - c0, time.process_time() - p0
        T._stage_timer.on = False
        return dt, cpu, pcpu, paths

    dt_plain, cpu_plain, _, plain = timed_loop("plain", False)
    dt, cpu, pcpu, cached = timed_loop("cached", True)
    stages = {k: round(v * 1e3 / n, 3) for k, v in T._stage_timer.host.items()}
    gpu_ms = T._stage_timer.gpu_ms() / n
    same = all(open(a, "rb").read() == open(b, "rb").read() for a, b in zip(cached, plain))
    ch, cw = Image.open(cached[0]).size[::-1]
    result = {"metric": f"adain_inference calls/s, the reference's {'video' if mode == 'video' else 'guide-view'} caller loop unchanged "
                        "(file / PIL in, stylised JPEG file out)", "value": round(n / dt, 2), "unit": "calls/s", "n_gpus": 1, "steps": n,
              "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / n, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f32", "data": "synthetic",
              "config": {"workload": (f"{n} calls: {h}x{w} JPEG frame files, content_size=256 -> {ch}x{cw}, use_depth=True with a proximity map per frame, "
                                      "one 933x700 style file resized to 512 (video/utils.py:341-350)" if mode == "video" else
                                      f"{n} calls: {h}x{w} PIL views, content_size=512 -> {ch}x{cw}, mask = view > 0, one 933x700 PIL style resized to 512 "
                                      "(Style_3DGS/train.py:86-115)") + ", fp32, seeded synthetic weights", "parallelism": "single GPU, one call at a time",
                         "schedule": args.schedule},
              "per_call": {"ms": round(dt * 1e3 / n, 3), "stages_ms": stages, "kernels_ms_by_hip_events": round(gpu_ms, 3),     # from the frame's H2D copy to the last kernel (upload, resize, ~27 launches)
                           "outside_the_stages_ms": round(dt * 1e3 / n - sum(stages.values()), 3),      # the caller's own work (PIL object / mask
                                                                                                       # construction in the loop) + bookkeeping
                           "calling_thread_cpu_ms": round(cpu * 1e3 / n, 3), "process_cpu_ms": round(pcpu * 1e3 / n, 3),
                           "style_encodes": 1, "stylised_mpixels_per_s": round(n * ch * cw / 1e6 / dt, 2)},
              "call_by_call_path": {"what": "set_style_cache(False): the style image is opened, resized and encoded in every call and the frame goes "
                                            "through the separate C-ABI calls, as the reference does (test.py:190-247)", "ms": round(dt_plain * 1e3 / n, 3),
                                    "calling_thread_cpu_ms": round(cpu_plain * 1e3 / n, 3), "files_identical_to_cached_path": bool(same)},
              "speedup_vs_call_by_call": round(dt_plain / dt, 3)}
    if not args.no_cpu:
        from oracle import adain_oracle as O

        torch.set_num_threads(int(os.environ.get("ADAIN_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16))))
        ct = T.test_transform(csize, False)(Image.open(os.path.join(root, "frames", "frame_0000.jpg")) if mode == "video" else Image.fromarray(frames[0])).unsqueeze(0)
        st = T.test_transform(512, False)(style_obj).unsqueeze(0)
        times = []
        with torch.no_grad():
            for _ in range(3):
                t0 = time.perf_counter()
                if mode == "video":
                    ref = O.quantize_u8(O.style_transfer(vgg_sd, dec_sd, ct, st, depth[0], 0.5, 0.30, 20))
                else:
                    ref = O.quantize_u8(O.mask_composite(ct, O.style_transfer_simple(vgg_sd, dec_sd, ct, st, 0.5), torch.from_numpy(frames[0].transpose(2, 0, 1) > 0)))
                times.append(time.perf_counter() - t0)
        result["cpu_baseline"] = {"value": round(1.0 / min(times), 3), "unit": "calls/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": f"3 forwards of call 0 through the oracle (tensor in, uint8 out; no file I/O), best {min(times) * 1e3:.0f} ms"}
    import shutil

    shutil.rmtree(root, ignore_errors=True)
    emit(result)
    ctx.finish()


_RESULT_OUT = None


def claim_stdout():
    """stdout carries ONE JSON line.  Native libraries write to file descriptor 1 on their own (RCCL's version banner - NCCL_DEBUG=VERSION
    is exported on this pool -, gloo's connection notes): from here on descriptor 1 IS stderr, and the result line goes to a private
    duplicate of the original stdout (``emit``)."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(result):
    out = _RESULT_OUT or sys.stdout
    out.write(json.dumps(result) + "\n")
    out.flush()


def main():
    global WEIGHT_SET
    args = parse_args()
    WEIGHT_SET = args.weights
    claim_stdout()
    ctx = Ctx(args)
    if args.per_call:
        return main_per_call(args, ctx)
    if args.job:
        return main_job(args, ctx)
    world, rank, device, use_dist, transport, shared_gpu = ctx.world, ctx.rank, ctx.device, ctx.use_dist, ctx.transport, ctx.shared_gpu
    barrier = ctx.barrier
    phases, t_phase = {}, [_T0]

    def phase(name):          # wall seconds of this process since the previous mark (the line says where the run's time went)
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - t_phase[0], 3)
        t_phase[0] = now

    phase("imports + process group")
    step = Step(device, config=args.config, first_frame=rank * args.batch, size=args.size, style_size=args.style_size, batch=args.batch, depth=args.depth,
                alpha=args.alpha)
    h, w, hs, ws = step.h, step.w, step.hs, step.ws
    n_job = world * args.batch                        # frames of one step's job over all ranks

    need_u8 = use_dist or args.config in (4, 5)
    oh, ow = (8 * step.hc, 8 * step.wc) if args.config != 5 else (h, w)     # config 5 composites at the view's own size

    def one_step(slot):
        # every config: the C-ABI call sequence of one forward on the step's resident frames (configs 4 / 5 end with their uint8
        # quantiser / mask composite; with more than one rank every config quantises - the gather moves uint8 frames); whole jobs
        # through the sharded driver are `--job`
        last[0] = step.run(to_u8=use_dist, u8_out=slot if need_u8 else None)

    last = [None]
    ctx.warm_transport(n_job, args.batch)
    telemetry = GpuTelemetry(device.index).start()
    phase("setup (weights, inputs, packing, transport)")
    dt, gathered, tinfo = jobs.run_timed_steps(one_step, args.steps, args.warmup, barrier=barrier, block_shape=(args.batch, oh, ow, 3),
                                               device=device, mode=args.gather, gather=use_dist, mark=step.engine.mark,
                                               elapsed=step.engine.elapsed)
    phase("warm-up + timed steps")
    telemetry.window("timed_region", tinfo["t0"], tinfo["t1"])
    out = last[0]
    if use_dist and rank == 0:
        expect = n_job * (args.steps if args.gather == "end" else 1)
        assert gathered is not None and gathered.shape[0] == expect, (None if gathered is None else gathered.shape, expect)
    del gathered

    # one isolated gather of one step's frames (nothing else in flight) for the transport's own cost
    gather_ms = None
    if use_dist:
        barrier()
        g0 = time.perf_counter()
        sh.gather_frames(step.u8, n_job, dst=0, counts=[args.batch] * world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    sustained = None
    if not use_dist and args.sustain > 0:
        # the same loop, run for at least --sustain seconds straight after the timed region: the headline's K steps last well
        # under a second at config 2, this says what the chip holds once clocks and temperature have settled
        per = max(dt / args.steps, 1e-4)
        k_s = int(math.ceil(args.sustain / per))
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(k_s):
            one_step(None)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - s0
        sustained = {"seconds": round(sdt, 3), "steps": k_s, "value": round(n_job * h * w / 1e6 / (sdt / k_s), 3),
                     "ms_per_step": round(sdt / k_s * 1e3, 4), "unit": "Mpixels/s"}
        telemetry.window("sustained_leg", s0, s0 + sdt)

    phase("isolated gather + sustained leg")
    # every rank's line of the table: HIP-event times of its own timed region and what its GPU's clock and power were meanwhile
    per_rank = [None] * world
    mine = {"rank": rank, "compute_ms": round(tinfo["compute_ms"], 3), "gather_ms": round(tinfo["gather_ms"], 3),
            "wall_ms": round(tinfo["local_s"] * 1e3, 3), "gpu": telemetry.stop()}
    if use_dist:
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]
    if rank != 0:
        # what follows on rank 0 (instrumented roofline leg, pixel-kernel table, the CPU baseline) needs nobody else: the other ranks
        # free their memory and wait on a store key, not in a collective (Ctx.finish)
        del step, last, out
        ctx.others_wait_for_rank0(args.launch_timeout)
        ctx.finish()
        return
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = n_job * h * w / 1e6 / (dt / args.steps)
        roof, layers, secondary = measure_roofline(step, 5)
        phase("roofline leg (instrumented steps)")
        result = base_result(args, ctx, value, ms, WORKLOADS[args.config].format(h=h, w=w, hs=hs, ws=ws, b=args.batch, alpha=args.alpha)
                             + ("; depth-aware variant (a proximity map per frame, offset 0.30, prominence 20)" if args.depth else ""),
                             f"frame sharding x{world}: contiguous frame blocks per rank, replicated weights and style statistics, "
                             + ("ONE gather of the uint8 frames of all timed steps to rank 0 at the end of the timed region" if args.gather == "end"
                                else "one gather of the uint8 frames to rank 0 per step (asynchronous)") if use_dist else "single GPU, no collective", "weak")
        result["roofline"] = roof
        result["secondary"] = secondary
        result["step_tflops"] = round(step.flops_per_step() / (dt / args.steps) / 1e12 * world, 2)
        result["ranks"] = ctx.ranks
        result["per_rank"] = per_rank
        if sustained is not None:
            sustained["ratio_to_value"] = round(sustained["value"] / value, 4)
            sustained["note"] = ("after idling the chip needs about 8 steps (30 ms of load) to reach its sustained clock: with few warm-up steps the K timed "
                                 "steps still contain the ramp (profiles/r04_step_ramp_after_idle.json, DESIGN.md section 6)")
            result["sustained"] = sustained
        if use_dist:
            result["gather"] = {"mode": args.gather, "in_timed_region": True, "transport": transport, "gathers_in_timed_region": tinfo["gathers"],
                                "bytes_per_rank_in_timed_region": int(tinfo["gather_bytes"]), "isolated_one_step_ms": round(gather_ms, 3),
                                "what": ("ONE gather of every rank's frames of all timed steps ends the timed region" if args.gather == "end"
                                         else "one asynchronous gather per step, at most two in flight, overlapping the next step's compute")}
            if args.rehearse:
                result["rehearsal"] = {"ranks_share_a_gpu": shared_gpu, "note": "not a multi-GPU measurement"}
        if args.pcie:
            # host buffers at the boundary: pinned fp32 frame in, pinned uint8 frame out, copies on the same stream
            host_in = step.content.cpu().pin_memory()
            host_out = torch.empty((args.batch, 8 * step.hc, 8 * step.wc, 3), dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            for _ in range(args.steps):
                step.content.copy_(host_in, non_blocking=True)
                o = step.run()
                host_out.copy_(rt.quantize_u8(o), non_blocking=True)
            torch.cuda.synchronize()
            pdt = (time.perf_counter() - p0) / args.steps
            result["pcie_inclusive"] = {"value": round(args.batch * h * w / 1e6 / pdt, 3), "unit": "Mpixels/s",
                                        "ms_per_step": round(pdt * 1e3, 4), "what": "pinned fp32 frame H2D + forward + uint8 D2H per step"}
        if not args.no_secondary:              # rank 0's GPU, after the timed region (the other ranks wait on the store key)
            result["secondary"] += measure_pixel_kernels(device)
        phase("pcie / secondary pixel kernels")
        if not args.no_cpu:                    # rank 0's host cores, after the timed region, at every N (north_star: "in the same run")
            out = step.run()
            torch.cuda.synchronize()
            cb, psnr, rel = cpu_baseline(step, step.u8 if args.config in (4, 5) else out)
            phase("cpu_baseline (the GPU idles)")
            if world > 1:
                cb["sample"] += f"; measured on rank 0 after the timed region (the other {world - 1} rank(s) have freed their device memory and wait on a key of the rendezvous store, in no collective)"
            result["cpu_baseline"] = cb
            result["psnr_db_vs_cpu"] = round(psnr, 2) if psnr != float("inf") else "inf"
            result["rel_l2_vs_cpu"] = float(f"{rel:.3e}")
        if args.layers and layers:
            for i, L in enumerate(layers):
                print(f"layer {i:2d}: {L['gflop']:8.2f} GF  {L['ms']:8.4f} ms  {L['tflops']:7.2f} TF/s", file=sys.stderr)
        # where this process's wall time went: the K timed steps are a fraction of a second of a run dominated by set-up and the CPU
        # baseline, so a utilisation sampler beside the whole run sees a mostly idle GPU
        result["run_phases_s"] = phases
        emit(result)
    ctx.finish(rank0_done=True)           # (the other ranks wait on the store key since the per-rank table)


if __name__ == "__main__":
    main()
