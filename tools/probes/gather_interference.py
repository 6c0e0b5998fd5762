#!/usr/bin/env python3
"""What a per-step RCCL gather costs the compute kernels on the rank that receives it - priced on ONE GPU.

A box has one GPU, so no peer can send anything; what can be measured is the part that decides the default of
``bench.py --gather``: an RCCL kernel issued every step on the communicator's own stream BESIDE the persistent, chip-filling
conv kernels of the next step (mode "overlap"), against the same bytes moved ONCE after the last step (mode "end").  A
world-size-1 ``nccl`` process group runs here (RANK=0, WORLD_SIZE=1); every gather moves ``--peers`` (7) dummy uint8 frames of the
step's size - the volume rank 0 of an 8-GPU node receives per step - through RCCL's gather (with one rank: RCCL's own device copy
kernel / copy engine on its stream, ordered against the compute stream by events exactly as a real gather is).

    python tools/probes/gather_interference.py [--steps 40] [--reps 5] [--peers 7]

Prints one JSON object: ms per step without any gather, with the overlapped per-step gather, and with the one end-of-region gather
(its cost spread over the steps), each the median of ``--reps`` timed regions."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")

import torch
import torch.distributed as dist

import bench
import applied_image_processing_amd.sharding as sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--peers", type=int, default=7)
    ap.add_argument("--config", type=int, default=2)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    step = bench.Step(dev, config=a.config)
    h, w = 8 * step.hc, 8 * step.wc
    dummy = torch.zeros((a.peers, h, w, 3), dtype=torch.uint8, device=dev)
    keep = torch.zeros((a.steps * a.peers, h, w, 3), dtype=torch.uint8, device=dev)
    u8 = torch.empty((1, h, w, 3), dtype=torch.uint8, device=dev)

    def region(mode):
        pending = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step.run(to_u8=True, u8_out=u8)
            if mode == "overlap":
                pending.append(sh.gather_frames(dummy, a.peers, dst=0, async_op=True, counts=[a.peers]))
                if len(pending) > 2:
                    pending.pop(0)()
        while pending:
            pending.pop(0)()
        if mode == "end":
            sh.gather_frames(keep, keep.shape[0], dst=0, counts=[keep.shape[0]])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    out = {"what": __doc__.split("\n")[0], "steps": a.steps, "reps": a.reps, "config": a.config,
           "bytes_per_step_gathered": int(dummy.numel()), "transport": sh.device_transport(dummy)}
    for mode in ("none", "overlap", "end"):          # warm every shape (communicator set-up, receive buffers) outside the timing
        region(mode)
    rows = {m: [] for m in ("none", "overlap", "end")}
    for _ in range(a.reps):                          # interleaved, so that clock drift hits every mode alike
        for m in rows:
            rows[m].append(region(m))
    for m, v in rows.items():
        out[f"ms_per_step_{m}"] = round(statistics.median(v), 4)
        out[f"ms_per_step_{m}_all"] = [round(x, 4) for x in v]
    out["overlap_vs_none"] = round(out["ms_per_step_overlap"] / out["ms_per_step_none"], 4)
    out["end_vs_none"] = round(out["ms_per_step_end"] / out["ms_per_step_none"], 4)
    # per-layer view of one overlapped step: does any conv launch stretch beside the transport kernel?
    for mode in ("none", "overlap"):
        per = [0.0] * 16
        for _ in range(5):
            fin = sh.gather_frames(dummy, a.peers, dst=0, async_op=True, counts=[a.peers]) if mode == "overlap" else None
            _, ev = step.run(timed=True)
            if fin is not None:
                fin()
            torch.cuda.synchronize()
            per = [x + e0.elapsed_time(e1) / 5 for x, (e0, e1) in zip(per, ev)]
        out[f"conv_layer_ms_{mode}"] = [round(x, 4) for x in per]
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
