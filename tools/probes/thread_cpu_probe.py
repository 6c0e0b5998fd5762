#!/usr/bin/env python3
"""Where does the CPU time of a job go, thread by thread?  Runs the 1080p video job (device-resident frames, then host-resident
frames) through ``jobs.stylize_frames_sharded`` and reads /proc/self/task/*/stat around it: CPU seconds per thread (named threads of
this package, the Python main thread, and whatever the HIP runtime starts), per frame.  Eight ranks share one host: this is what
each of them costs it.

    python tools/probes/thread_cpu_probe.py [--frames 96] [--env HIP_FORCE_... (none needed)]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch

import bench
import applied_image_processing_amd.engine as engine_mod
import applied_image_processing_amd.jobs as jobs
import applied_image_processing_amd.synth as synth

TICK = os.sysconf("SC_CLK_TCK")


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            s = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        name = s[s.index("(") + 1:s.rindex(")")]
        f = s[s.rindex(")") + 2:].split()
        out[int(tid)] = (name, (int(f[11]) + int(f[12])) / TICK)          # utime + stime
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--resident-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    born = {"before HIP": set(threads())}
    torch.cuda.set_device(0)
    torch.zeros(1, device=dev)
    born["HIP initialisation"] = set(threads())
    w = bench.synth_weights()
    eng = engine_mod.AdaINEngine(w[0], w[1], dev)
    born["engine (first kernels)"] = set(threads())
    style = torch.from_numpy(synth.image(4, 1, 512, 512)).to(dev)
    cache = {}
    res = {}
    for host in ((False,) if a.resident_only else (False, True)):
        store = bench.FrameStore(4, a.frames, 0, a.frames, 1080, 1920, dev, host=host)
        jobs.stylize_frames_sharded(eng, store, style, style_cache=cache, gather=False)          # warm
        torch.cuda.synchronize()
        born.setdefault("first job", set(threads()))
        t0, w0, p0 = threads(), time.perf_counter(), time.process_time()
        _, info = jobs.stylize_frames_sharded(eng, store, style, style_cache=cache, gather=False)
        torch.cuda.synchronize()
        t1, wall, proc = threads(), time.perf_counter() - w0, time.process_time() - p0
        rows, top = {}, []
        for tid, (name, cpu) in t1.items():
            d = cpu - t0.get(tid, (name, 0.0))[1]
            key = "MainThread" if tid == os.getpid() else name
            rows[key] = rows.get(key, 0.0) + d
            phase = next((ph for ph, ids in born.items() if tid in ids), "this job")
            top.append((round(d * 1e3 / a.frames, 3), key, tid - os.getpid(), "born: " + phase))
        res["host_frames" if host else "resident_frames"] = {
            "wall_s": round(wall, 3), "process_cpu_s": round(proc, 3), "launch_thread_cpu_s": round(info["host_cpu_s"], 3),
            "cpu_ms_per_frame_by_thread": {k: round(v * 1e3 / a.frames, 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]) if v > 0},
            "busiest_threads_ms_per_frame": sorted(top, reverse=True)[:4], "threads": len(t1), "feeder": info["feeder"],
            "cpu_of_threads_that_ended_with_the_job_s": round(proc - sum(rows.values()), 3)}
        del store
    # the same question for a bare kernel loop (no feeder, no second stream, no events): config-2 steps for about half a second
    step = bench.Step(dev, config=2, engine=eng, weights=w)
    step.run()
    torch.cuda.synchronize()
    t0, w0, p0 = threads(), time.perf_counter(), time.process_time()
    for _ in range(150):
        step.run()
    enq = time.perf_counter() - w0
    torch.cuda.synchronize()
    t1, wall, proc = threads(), time.perf_counter() - w0, time.process_time() - p0
    top = sorted(((round(cpu - t0.get(tid, (n_, 0.0))[1], 3), "MainThread" if tid == os.getpid() else n_, tid - os.getpid())
                  for tid, (n_, cpu) in t1.items()), reverse=True)[:3]
    res["bare_step_loop"] = {"wall_s": round(wall, 3), "enqueue_s": round(enq, 3), "process_cpu_s": round(proc, 3), "busiest_threads_cpu_s": top}
    res["env"] = {k: v for k, v in os.environ.items() if k.split("_")[0] in ("HSA", "HIP", "ROC", "ROCR", "AMD", "GPU", "NCCL", "RCCL")}
    print(json.dumps({"frames": a.frames, "what": "CPU per thread of a 1080p job", **res}), flush=True)


if __name__ == "__main__":
    main()
