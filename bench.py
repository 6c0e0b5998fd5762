#!/usr/bin/env python3
"""Throughput benchmark of the AdaIN hot path on MI355X (BASELINE.json metric: stylised Mpixels/s).

One "step" = one full ``style_transfer_simple`` forward (reference Style_3DGS/AdaIN/test.py:74-81) on
one synthetic batch that is already resident in HBM: encode content (1024x1024, batch 1) + encode
style (512x512) + channel statistics + AdaIN/alpha blend + decode.  The style is re-encoded every
step, as the reference does on every call; nothing is cached across steps.  fp32 throughout.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``:
one process per GPU, every rank runs the same per-GPU workload on its own frames (weak scaling, no
data-path collective); the timed region is bracketed by barrier + synchronize and the MAX over ranks is
reported.  After the timed region the uint8 frames are gathered once to rank 0 over RCCL ("final gather").

Rank 0 prints ONE JSON line.  ``roofline`` is measured live with HIP events recorded by the C ABI on the
launch stream around every 3x3-conv launch (the dominant kernel family, fp32 MFMA); ``cpu_baseline`` is
the CPU oracle (a torch-CPU restatement of the reference path) timed on this node's host cores on the
same workload.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch
import torch.distributed as dist

import applied_image_processing_amd.arch as arch
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.sharding as sh
import applied_image_processing_amd.synth as synth

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)"


def conv3x3_layer_flops(h, w, hs, ws):
    """Algorithmic flops (2 per MAC) of the generic 3x3 conv launches of one step, in launch order:
    8 encoder convs for the content image, 8 for the style image, 8 decoder convs."""
    def enc(hh, ww):
        out = []
        for L in arch.encoder_plan()[2:]:
            if L["src"] == "pool":
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
            out.append(2 * hh * ww * L["cin"] * L["cout"] * 9)
        return out

    def dec(hh, ww):
        out = []
        for L in arch.decoder_plan()[:-1]:
            if L["src"] == "up":
                hh, ww = 2 * hh, 2 * ww
            out.append(2 * hh * ww * L["cin"] * L["cout"] * 9)
        return out

    hc, wc = arch.encoded_size(h, w)
    return enc(h, w), enc(hs, ws), dec(hc, wc)


class Step:
    """style_transfer_simple on raw device buffers through the C ABI."""

    def __init__(self, device, seed_offset=0, h=1024, w=1024, hs=512, ws=512, batch=1, alpha=0.5):
        self.alpha = alpha
        vgg_sd = synth.to_torch(synth.vgg_state_dict(0, full=False))
        dec_sd = synth.to_torch(synth.decoder_state_dict(0))
        self.enc = rt.pack_encoder(vgg_sd, device)
        self.dec = rt.pack_decoder(dec_sd, device)
        self.content = torch.from_numpy(synth.image(3 + 1000 * seed_offset, batch, h, w)).to(device)
        self.style = torch.from_numpy(synth.image(4, 1, hs, ws)).to(device)

    def run(self, ev_c=None, ev_s=None, ev_d=None):
        cf = rt.encode(self.content, self.enc, ev_c)
        sf = rt.encode(self.style, self.enc, ev_s)
        c_mean, c_std = rt.mean_std(cf, True)
        s_mean, s_std = rt.mean_std(sf, True)
        g = rt.blend_alpha(cf, True, c_mean, c_std, s_mean, s_std, self.alpha)
        return rt.decode(g, self.dec, ev_d)


def make_events(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in evs:
        e.record()          # creates the underlying hipEvent_t; the C ABI re-records it
    return evs


def measure_roofline(step, reps, h, w, hs, ws):
    fl_c, fl_s, fl_d = conv3x3_layer_flops(h, w, hs, ws)
    flops = fl_c + fl_s + fl_d
    total_ms = 0.0
    per_layer = [0.0] * len(flops)
    for _ in range(reps):
        ev_c, ev_s, ev_d = make_events(11), make_events(11), make_events(10)
        step.run(ev_c, ev_s, ev_d)
        torch.cuda.synchronize()
        # encode: events 1..9 bracket the 8 generic convs; decode: events 0..8
        d = [ev_c[i + 1].elapsed_time(ev_c[i + 2]) for i in range(8)]
        d += [ev_s[i + 1].elapsed_time(ev_s[i + 2]) for i in range(8)]
        d += [ev_d[i].elapsed_time(ev_d[i + 1]) for i in range(8)]
        per_layer = [a + b for a, b in zip(per_layer, d)]
        total_ms += sum(d)
    launches = len(flops) * reps
    avg_ms = total_ms / launches
    achieved = sum(flops) * reps / (total_ms * 1e-3) / 1e12
    layers = [{"gflop": f / 1e9, "ms": t / reps, "tflops": f / (t / reps * 1e-3) / 1e12} for f, t in zip(flops, per_layer)]
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
        "kernel": "conv3x3_mfma_kernel (24 launches/step)", "avg_launch_ms": round(avg_ms, 4),
        "flop_per_launch_avg": sum(flops) / len(flops),
    }, layers


def cpu_baseline(h, w, hs, ws, gpu_out):
    from oracle import adain_oracle as O

    # threads actually used: the GPU box grants a 1-GPU job a CPU share of 16 cores (more threads than that
    # oversubscribe and run several times slower); ADAIN_CPU_THREADS overrides
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("ADAIN_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    vgg_sd = synth.to_torch(synth.vgg_state_dict(0, full=False))
    dec_sd = synth.to_torch(synth.decoder_state_dict(0))
    c = torch.from_numpy(synth.image(3, 1, h, w))
    s = torch.from_numpy(synth.image(4, 1, hs, ws))
    times = []
    with torch.no_grad():
        ref = O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5)          # warm-up, also the PSNR reference
        t_end = time.time() + 20.0
        while len(times) < 2 or (time.time() < t_end and len(times) < 5):
            t0 = time.perf_counter()
            O.style_transfer_simple(vgg_sd, dec_sd, c, s, 0.5)
            times.append(time.perf_counter() - t0)
    best = min(times)
    out = gpu_out[:1].cpu()
    rel = float((out - ref).norm() / ref.norm())
    psnr = float(O.psnr(out.clamp(0, 1), ref.clamp(0, 1)).min())
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {
        "value": round(h * w / 1e6 / best, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"{len(times)} full forwards of the same workload ({h}x{w} content + {hs}x{ws} style), best of "
                  f"{len(times)}; median {statistics.median(times):.2f} s; torch {torch.__version__} CPU, {model}",
    }, psnr, rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024, help="content H = W")
    ap.add_argument("--style-size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer table to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the AdaIN path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    h = w = args.size
    hs = ws = args.style_size
    step = Step(device, seed_offset=rank, h=h, w=w, hs=hs, ws=ws, batch=args.batch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step.run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    # final gather of the finished uint8 frames to rank 0 (outside the timed region)
    gather_ms = None
    u8 = rt.quantize_u8(out)
    if world > 1:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        allf = sh.gather_frames(u8, world * args.batch, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert allf.shape[0] == world * args.batch

    result = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * h * w / 1e6 / (dt / args.steps)
        roof, layers = measure_roofline(step, 3, h, w, hs, ws) if args.batch == 1 else (None, None)
        result = {
            "metric": "stylized Mpixels/sec, AdaIN forward (encode content + encode style + AdaIN + decode)",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {h}x{w} AdaIN forward, batch={args.batch} per GPU, fp32, style {hs}x{ws} "
                                   "re-encoded every step, alpha=0.5, seeded synthetic weights (reference architecture)",
                       "parallelism": f"frame sharding x{world}, no data-path collective"},
            "roofline": roof,
        }
        flop_step = (arch.conv_flops_encoder(h, w) + arch.conv_flops_encoder(hs, ws)
                     + arch.conv_flops_decoder(*arch.encoded_size(h, w))) * args.batch
        result["step_tflops"] = round(flop_step / (dt / args.steps) / 1e12 * world, 2)
        if gather_ms is not None:
            result["final_gather_ms"] = round(gather_ms, 3)
        if world == 1 and not args.no_cpu:
            cb, psnr, rel = cpu_baseline(h, w, hs, ws, out)
            result["cpu_baseline"] = cb
            result["psnr_db_vs_cpu"] = round(psnr, 2)
            result["rel_l2_vs_cpu"] = float(f"{rel:.3e}")
        if args.layers and layers:
            for i, L in enumerate(layers):
                print(f"layer {i:2d}: {L['gflop']:8.2f} GF  {L['ms']:8.4f} ms  {L['tflops']:7.2f} TF/s", file=sys.stderr)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
