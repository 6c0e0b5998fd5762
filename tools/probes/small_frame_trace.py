#!/usr/bin/env python3
"""One small frame at a time through the one-call entry point (adain_stylize_u8), the reference callers' operating point
(video/utils.py:261-270: content_size = 256 -> 256 x 456 frames; test.py:160: 512).

  python tools/probes/small_frame_trace.py run [H W [N]]      - the loop itself (run it under rocprofv3 --kernel-trace)
  python tools/probes/small_frame_trace.py report <dir>       - per-dispatch table of one call from the trace: kernel, grid, duration,
                                                                gap to the previous dispatch; and the per-call totals

The `run` leg also prints the wall-clock rate of back-to-back calls (frames/s) for sub-batches of 1 and N."""
import csv
import glob
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(h, w, nb, reps):
    if os.environ.get("ADAIN_PROBE_DIAG_LIB"):      # A/B of a diagnostic-library switch (e.g. ADAIN_W4_PERSIST_ROUNDS_X10)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _diag  # noqa: F401
    import torch
    import bench
    import applied_image_processing_amd.engine as engine_mod
    import applied_image_processing_amd.synth as synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wts = bench.synth_weights()
    eng = engine_mod.AdaINEngine(wts[0], wts[1], dev)
    eng.set_style(torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev))
    import applied_image_processing_amd.runtime as rt

    frames = torch.stack([synth.frame_u8_torch(7 + k, h, w, dev) for k in range(nb)])
    # A B A B on the same box: batch schedule, latency schedule (cin split of the under-filled layers), single frames; then the batch
    for sub, sched, name in ((1, rt.SCHEDULE_BATCH, "batch  "), (1, rt.SCHEDULE_LATENCY, "latency"), (1, rt.SCHEDULE_BATCH, "batch  "),
                             (1, rt.SCHEDULE_LATENCY, "latency"), (nb, rt.SCHEDULE_BATCH, "batch  ")):
        with rt.schedule(sched):
            out = None
            for _ in range(5):
                out = eng.stylize_u8(frames[:sub], alpha=0.5, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = eng.stylize_u8(frames[:sub], alpha=0.5, out=out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{h}x{w} sub-batch {sub:2d} schedule {name}: {reps * sub / dt:8.1f} frames/s  {dt / reps * 1e3:7.3f} ms per call "
              f"({dt / reps / sub * 1e3:.3f} ms per frame)", flush=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("adain::", "")[:70]


def is_call_start(k):
    return k.startswith("conv_first_kernel<true")


def report(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 0) or 1)))
    rows.sort()
    # calls start with conv_first_kernel<true> (the uint8 entry); take the sub-batch-1 calls = the first 5 + reps of them
    starts = [i for i, r in enumerate(rows) if is_call_start(r[2])]
    if not starts:
        print("no conv_first_kernel dispatch found")
        return
    calls = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
    # the single-frame legs: calls whose first kernel has the first call's grid; one table per dispatch count (= per schedule)
    g0 = calls[0][0][3]
    import statistics
    for n0 in sorted({len(c) for c in calls if c[0][3] == g0}):
        first = [c for c in calls if len(c) == n0 and c[0][3] == g0]
        if len(first) < 8:
            continue
        report_calls(first)


def report_calls(first):
    import statistics
    mid = first[len(first) // 2]         # a warm call in the middle
    print(f"## one call ({len(mid)} dispatches), from the middle of {len(first)} alike\n")
    print("| # | kernel | workgroups | us | gap before, us |")
    print("|---|---|---|---|---|")
    prev_end = None
    for i, (s, e, k, g, wg) in enumerate(mid):
        gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:.2f}"
        print(f"| {i} | `{k}` | {g // max(wg, 1)} | {(e - s) / 1e3:.2f} | {gap} |")
        prev_end = e
    busy = sum(e - s for s, e, *_ in mid) / 1e3
    span = (mid[-1][1] - mid[0][0]) / 1e3
    print(f"\nkernels {busy:.1f} us, first start to last end {span:.1f} us")
    # median over the alike calls
    spans = [(c[-1][1] - c[0][0]) / 1e3 for c in first[3:]]
    busys = [sum(e - s for s, e, *_ in c) / 1e3 for c in first[3:]]
    per = [(c2[0][0] - c1[0][0]) / 1e3 for c1, c2 in zip(first[3:], first[4:])]
    print(f"median over {len(spans)} calls: kernels {statistics.median(busys):.1f} us, span {statistics.median(spans):.1f} us, "
          f"start-to-start {statistics.median(per):.1f} us")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        a = sys.argv[2:]
        h = int(a[0]) if len(a) > 0 else 256
        w = int(a[1]) if len(a) > 1 else 456
        nb = int(a[2]) if len(a) > 2 else 26
        reps = int(a[3]) if len(a) > 3 else 200
        run(h, w, nb, reps)
