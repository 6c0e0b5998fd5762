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
WINOGRAD = os.environ.get("ADAIN_WINOGRAD", "1") != "0"      # the C library's default; ADAIN_WINOGRAD=0 = direct implicit GEMM
WINO_FORM = int(os.environ.get("ADAIN_WINO_MH", "5"))          # the C library's default: 5 = F(4,3) x F(2,3); others F(2x2,3x3)
# multiplies the conv3x3 kernel executes on the matrix pipe per direct-convolution multiply
EXECUTED = (24.0 / 72.0 if WINO_FORM == 5 else 16.0 / 36.0) if WINOGRAD else 1.0
CONV_KERNEL = ("conv3x3_mfma_kernel / conv3x3_persist_kernel" if not WINOGRAD else
               "conv3x3_wino4_kernel (Winograd F(4,3) x F(2,3))" if WINO_FORM == 5 else
               "conv3x3_wino2_kernel + conv3x3_wino3_kernel (Winograd F(2x2,3x3))")
WORKLOADS = {
    2: "configs[1]: {h}x{w} AdaIN forward (style_transfer_simple), batch={b} per GPU, style {hs}x{ws} re-encoded every step, alpha=0.5",
    3: "configs[2]: {h}x{w} depth-aware AdaIN (style_transfer, proximity-map blend), batch={b} per GPU, style {hs}x{ws} re-encoded every step",
    4: "configs[3]: video frames {h}x{w}, {b} frames per GPU per step, one {hs}x{ws} style (statistics cached for the job), uint8 out",
    5: "configs[4]: 3DGS guide views {h}x{w} with masks, {b} views per GPU per step, one {hs}x{ws} style, mask composite + uint8 out",
}


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


class Step:
    """One pass of the hot path on raw device buffers through the C ABI.  ``config``:
    2: style_transfer_simple, 1024x1024 content + 512x512 style re-encoded every step (test.py:74-81)
    3: style_transfer (depth-aware), 2048x2048 content + depth map + 512x512 style (test.py:52-71)
    4: video frames 1080x1920, `batch` frames per step, ONE style for the job (statistics computed once, before the
       timed region: a video has one style, video/utils.py:341), output quantised to uint8
    5: 3DGS guide views 1200x1600 with masks, `batch` views per step, one style, mask composite + uint8 (train.py:86-115)
    """

    def __init__(self, device, config=2, seed_offset=0, size=None, style_size=512, batch=1, alpha=0.5):
        self.config, self.alpha, self.batch = config, alpha, batch
        self.h, self.w = {2: (1024, 1024), 3: (2048, 2048), 4: (1080, 1920), 5: (1200, 1600)}[config]
        if size:
            self.h = self.w = size
        self.hs = self.ws = style_size
        vgg_sd = synth.to_torch(synth.vgg_state_dict(0, full=False))
        dec_sd = synth.to_torch(synth.decoder_state_dict(0))
        self.enc = rt.pack_encoder(vgg_sd, device)
        self.dec = rt.pack_decoder(dec_sd, device)
        base = {2: 3, 3: 5, 4: 7, 5: 1000}[config] + 1000 * seed_offset
        self.content = torch.cat([torch.from_numpy(synth.image(base + i, 1, self.h, self.w)) for i in range(batch)]).to(device)
        self.style = torch.from_numpy(synth.image(4, 1, self.hs, self.ws)).to(device)
        self.hc, self.wc = rt.encoded_size(self.h, self.w)
        self.style_each_step = config in (2, 3)
        if config == 3:
            self.depth = [torch.from_numpy(synth.smooth_depth(6 + i, self.h, self.w)).to(device) for i in range(batch)]
        if config == 5:
            bg = torch.cat([torch.from_numpy(synth.uniform01(2000 + i, self.h * self.w).reshape(1, 1, self.h, self.w) < 0.3)
                            for i in range(batch)]).to(device)
            self.content = torch.where(bg, torch.zeros_like(self.content), self.content)
            self.mask = (self.content > 0).float()
        if not self.style_each_step:
            sf = rt.encode(self.style, self.enc)
            self.s_mean, self.s_std = rt.mean_std(sf, True)

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
        if self.style_each_step:
            b += enc(1, self.hs, self.ws)
        return b + dec(self.batch, self.hc, self.wc)

    def conv3x3_flops(self):
        f = enc_conv3x3_flops(self.batch, self.h, self.w)
        if self.style_each_step:
            f += enc_conv3x3_flops(1, self.hs, self.ws)
        return f + dec_conv3x3_flops(self.batch, self.hc, self.wc)

    def run(self, timed=False):
        ev = []
        self.edge_ev = []
        ev_c = make_events(11) if timed else None
        cf = rt.encode(self.content, self.enc, ev_c)
        if timed:
            ev += [(ev_c[i + 1], ev_c[i + 2]) for i in range(8)]
            self.edge_ev.append(("conv_first(content)", ev_c[0], ev_c[1]))
        if self.style_each_step:
            ev_s = make_events(11) if timed else None
            sf = rt.encode(self.style, self.enc, ev_s)
            s_mean, s_std = rt.mean_std(sf, True)
            if timed:
                ev += [(ev_s[i + 1], ev_s[i + 2]) for i in range(8)]
                self.edge_ev.append(("conv_first(style)", ev_s[0], ev_s[1]))
        else:
            s_mean, s_std = self.s_mean, self.s_std
        c_mean, c_std = rt.mean_std(cf, True)
        if self.config == 3:
            p = torch.cat([rt.strength_map(d, self.hc, self.wc, 0.15, 20) for d in self.depth])
            g = rt.blend_pmap(cf, True, c_mean, c_std, s_mean, s_std, p)
        else:
            g = rt.blend_alpha(cf, True, c_mean, c_std, s_mean, s_std, self.alpha)
        ev_d = make_events(10) if timed else None
        out = rt.decode(g, self.dec, ev_d)
        if timed:
            ev += [(ev_d[i], ev_d[i + 1]) for i in range(8)]
            self.edge_ev.append(("conv_last", ev_d[8], ev_d[9]))
        if self.config == 5:
            size = (self.h, self.w)
            out = rt.mask_composite(self.content, rt.resize_bilinear(out, size), rt.resize_nearest(self.mask, size))
        if self.config in (4, 5):
            self.u8 = rt.quantize_u8(out)
        return (out, ev) if timed else out


def load_pmc_traffic(workload_key):
    """HBM bytes per conv3x3 launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/summarize_rocprof.py --traffic); None when no profile of this workload is committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d.get(workload_key, {}).get("hbm_bytes_per_conv3x3_launch")
    except Exception:
        return None


def measure_roofline(step, reps):
    flops = step.conv3x3_flops()
    total_ms = 0.0
    per_layer = [0.0] * len(flops)
    step.run()                       # back to steady state after the gather / host work in between
    step.run(timed=True)
    torch.cuda.synchronize()
    for _ in range(reps):
        _, ev = step.run(timed=True)
        torch.cuda.synchronize()
        d = [a.elapsed_time(b) for a, b in ev]
        step.edge_ms = {name: a.elapsed_time(b) for name, a, b in step.edge_ev}
        per_layer = [x + y for x, y in zip(per_layer, d)]
        total_ms += sum(d)
    launches = len(flops) * reps
    avg_ms = total_ms / launches
    achieved = sum(flops) * reps / (total_ms * 1e-3) / 1e12
    layers = [{"gflop": f / 1e9, "ms": t / reps, "tflops": f / (t / reps * 1e-3) / 1e12} for f, t in zip(flops, per_layer)]
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        "traffic": load_pmc_traffic(f"config{step.config}_batch{step.batch}"),
        "kernel": CONV_KERNEL + f" ({len(flops)} launches/step)",
        "avg_launch_ms": round(avg_ms, 4), "flop_per_launch_avg": sum(flops) / len(flops),
        # `achieved` counts the ALGORITHMIC flops of the direct 3x3 convolution.  The Winograd kernels execute
        # 24/72 (F(4,3) x F(2,3): 24 multiplies per 4 x 2 outputs) or 16/36 (F(2x2,3x3)) of those multiplies on the matrix
        # pipe, which is how `frac` can exceed 1.
        "executed_mfma_tflops": round(achieved * EXECUTED, 2),
        "executed_mfma_frac": round(achieved * EXECUTED / PEAK_FP32_MFMA_TFLOPS, 4),
        "algorithmic_bytes_per_launch_avg": sum(step.conv3x3_algorithmic_bytes()) / len(flops),
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
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] index + 1 (default 2 = configs[1])")
    ap.add_argument("--size", type=int, default=0, help="override content H = W")
    ap.add_argument("--style-size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--pcie", action="store_true", help="also report the rate with the frame crossing PCIe both ways (never `value`)")
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
    local_rank %= torch.cuda.device_count()           # one rank per GPU on a node; wraps only in single-GPU rehearsals
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ      # torch.distributed.run sets RANK even for one process
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # gloo carries the host-side rendezvous (barriers, the MAX of the step time); RCCL ("nccl") carries device
        # tensors, i.e. the final gather of the finished frames over xGMI
        dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)

    step = Step(device, config=args.config, seed_offset=rank, size=args.size, style_size=args.style_size, batch=args.batch)
    h, w, hs, ws = step.h, step.w, step.hs, step.ws

    def barrier():
        torch.cuda.synchronize()                      # this rank's GPU work is done ...
        if use_dist:
            dist.all_reduce(torch.zeros(1))           # ... and so is everybody else's (host rendezvous, gloo)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step.run()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    # final gather of the finished uint8 frames to rank 0 (outside the timed region)
    gather_ms, gather_via = None, None
    u8 = rt.quantize_u8(out)
    if use_dist:
        torch.cuda.synchronize()
        for via in ("rccl", "gloo"):
            try:
                g0 = time.perf_counter()
                allf = sh.gather_frames(u8 if via == "rccl" else u8.cpu(), world * args.batch, dst=0)
                torch.cuda.synchronize()
                gather_ms, gather_via = (time.perf_counter() - g0) * 1e3, via
                break
            except Exception as e:       # the harness must still report the compute numbers if RCCL cannot start
                print(f"[bench] final gather over {via} failed on rank {rank}: {e}", file=sys.stderr)
        if rank == 0 and gather_via:
            assert allf.shape[0] == world * args.batch

    result = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * h * w / 1e6 / (dt / args.steps)
        roof, layers = measure_roofline(step, 5)
        result = {
            "metric": "stylized Mpixels/sec, AdaIN forward (encode content + encode style + AdaIN + decode)",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config].format(h=h, w=w, hs=hs, ws=ws, b=args.batch)
                                   + ", fp32, seeded synthetic weights (reference architecture)",
                       "parallelism": f"frame sharding x{world}, no data-path collective"},
            "roofline": roof,
        }
        result["step_tflops"] = round(step.flops_per_step() / (dt / args.steps) / 1e12 * world, 2)
        if gather_ms is not None:
            result["final_gather_ms"] = round(gather_ms, 3)
            result["final_gather_transport"] = gather_via
        if args.pcie:
            # host buffers at the boundary: pinned fp32 frame in, pinned uint8 frame out, copies on the same stream
            host_in = step.content.cpu().pin_memory()
            host_out = torch.empty((args.batch, out.shape[2], out.shape[3], 3), dtype=torch.uint8).pin_memory()
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
        if world == 1 and not args.no_cpu and args.config == 2 and args.batch == 1:
            cb, psnr, rel = cpu_baseline(h, w, hs, ws, out)
            result["cpu_baseline"] = cb
            result["psnr_db_vs_cpu"] = round(psnr, 2)
            result["rel_l2_vs_cpu"] = float(f"{rel:.3e}")
        if args.layers and layers:
            for i, L in enumerate(layers):
                print(f"layer {i:2d}: {L['gflop']:8.2f} GF  {L['ms']:8.4f} ms  {L['tflops']:7.2f} TF/s", file=sys.stderr)
            for name, ms in getattr(step, "edge_ms", {}).items():
                print(f"{name}: {ms:.4f} ms", file=sys.stderr)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
