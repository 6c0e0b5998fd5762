#!/usr/bin/env python3
"""Which threads of a process burn CPU while the GPU runs a bare kernel loop, and which wait primitive the host uses: prints CPU
seconds per thread for ~0.5 s of config-2 steps followed by (a) torch.cuda.synchronize(), (b) Event.synchronize(), (c)
Event(blocking=True).synchronize(), (d) polling Event.query() with a 0.2 ms sleep.  Run it under different environments (HSA_*,
ROC_*, AMD_*) to see whether the ROCr / ROCclr service thread can be made to sleep:  python tools/probes/spin_probe.py <label>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

TICK = os.sysconf("SC_CLK_TCK")


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            s = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        f = s[s.rindex(")") + 2:].split()
        out[int(tid)] = (int(f[11]) + int(f[12])) / TICK
    return out


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
step = bench.Step(dev, config=2)
for _ in range(3):
    step.run()
torch.cuda.synchronize()
res = {"label": sys.argv[1] if len(sys.argv) > 1 else "", "env": {k: v for k, v in os.environ.items() if k.split("_")[0] in ("HSA", "ROC", "AMD", "GPU")}}
for how in ("device_synchronize", "event_synchronize", "blocking_event_synchronize", "query_sleep"):
    t0, w0 = threads(), time.perf_counter()
    for _ in range(60):
        step.run()
    ev = torch.cuda.Event(blocking=(how == "blocking_event_synchronize"))
    ev.record()
    enq = time.perf_counter() - w0
    if how == "device_synchronize":
        torch.cuda.synchronize()
    elif how == "query_sleep":
        while not ev.query():
            time.sleep(2e-4)
    else:
        ev.synchronize()
    wall = time.perf_counter() - w0
    t1 = threads()
    main = t1[os.getpid()] - t0[os.getpid()]
    others = sorted((round(c - t0.get(t, 0.0), 3) for t, c in t1.items() if t != os.getpid()), reverse=True)[:2]
    res[how] = {"wall_s": round(wall, 3), "enqueue_s": round(enq, 3), "main_thread_cpu_s": round(main, 3), "busiest_other_threads_cpu_s": others}
print(json.dumps(res), flush=True)
