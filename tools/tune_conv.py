#!/usr/bin/env python3
"""Times the 3x3-conv tile variants on the layer shapes of a workload (interleaved rounds in one process,
median of N; cdna_hip_programming.md rule 24).  Usage on the GPU box:
    python tools/tune_conv.py [--size 1024] [--rounds 7] [--variants 0,1,2,3,4,5]
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401,E402  (selects libadain_hip_diag.so)
import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth


def layers_for(size):
    s = size
    L = [  # name, cin, cout, h (conv output, before output pool), mode, pool_out
        ("conv1_2", 64, 64, s, rt.SRC_DIRECT, True),
        ("conv2_1", 64, 128, s // 2, rt.SRC_DIRECT, False),
        ("conv2_2", 128, 128, s // 2, rt.SRC_DIRECT, True),
        ("conv3_1", 128, 256, s // 4, rt.SRC_DIRECT, False),
        ("conv3_2", 256, 256, s // 4, rt.SRC_DIRECT, False),
        ("conv3_4p", 256, 256, s // 4, rt.SRC_DIRECT, True),
        ("conv4_1", 256, 512, s // 8, rt.SRC_DIRECT, False),
        ("dec1", 512, 256, s // 8, rt.SRC_DIRECT, False),
        ("dec2_up", 256, 256, s // 4, rt.SRC_UP2X, False),
        ("dec5", 256, 128, s // 4, rt.SRC_DIRECT, False),
        ("dec6_up", 128, 128, s // 2, rt.SRC_UP2X, False),
        ("dec7", 128, 64, s // 2, rt.SRC_DIRECT, False),
        ("dec8_up", 64, 64, s, rt.SRC_UP2X, False),
    ]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", type=str, default="0,1,2,3,4")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--burst", type=int, default=10)
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda", 0)
    print(f"{'layer':10s} {'shape':>22s} " + " ".join(f"{'v%d TF/s' % v:>9s}" for v in variants))
    for name, cin, cout, h, mode, pool in layers_for(args.size):
        if args.only and name not in args.only.split(","):
            continue
        hs = h // 2 if mode == rt.SRC_UP2X else h
        x = torch.from_numpy(synth.uniform_sym(1, (1, hs, hs, cin), 1.0)).to(dev)
        w = torch.from_numpy(synth.uniform_sym(2, (cout, cin, 3, 3), (6.0 / (9 * cin)) ** 0.5)).to(dev)
        b = torch.zeros(cout, device=dev)
        packed = rt.conv3x3_pack(w)
        wino = rt.conv3x3_wino_pack(w) if cin % 8 == 0 and mode != rt.SRC_POOL2 else None
        wino4 = rt.conv3x3_wino_pack(w, 5) if 24 in variants else None

        def run(v):
            if v in (20, 21, 22, 23, 24):      # Winograd F(2x2,3x3): 20 = 8 waves (2 M-tiles), 21 = 4 waves (1 M-tile)
                if wino is None:
                    raise rt.AdainHipError("no winograd form")
                if v == 24:
                    return rt.conv3x3_wino(x, wino4, b, cout, mode, True, pool, 5)
                return rt.conv3x3_wino(x, wino, b, cout, mode, True, pool, {20: 2, 21: 1, 22: 3, 23: 4, 24: 5}[v])
            return rt.conv3x3(x, packed, b, cout, mode, True, pool, v)

        flops = 2.0 * h * h * cin * cout * 9
        times = {v: [] for v in variants}
        ok = {}
        for v in variants:
            try:
                run(v)
                ok[v] = True
            except rt.AdainHipError:
                ok[v] = False
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in variants:
                if not ok[v]:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.burst):          # back-to-back launches: steady-state clocks, no idle gaps
                    run(v)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.burst)
        cells = []
        for v in variants:
            if ok[v]:
                t = statistics.median(times[v])
                cells.append(f"{flops / (t * 1e-3) / 1e12:9.1f}")
            else:
                cells.append(f"{'-':>9s}")
        shape = f"{cin}->{cout} @{h} m{mode}{'p' if pool else ''}"
        print(f"{name:10s} {shape:>22s} " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
