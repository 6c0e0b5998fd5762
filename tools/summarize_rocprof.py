#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace and/or PMC counter collection) into a small markdown
table per kernel name: calls, total / average duration, share; and per-kernel counter averages.

    python tools/summarize_rocprof.py <rocprof output dir> > profiles/<name>.md
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("adain::", "")
    return name[:90]


def main(d):
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if traces:
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        for f in traces:
            for r in csv.DictReader(open(f)):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
        total = sum(a[1] for a in agg.values())
        print("## kernel trace (durations in microseconds)\n")
        print("| kernel | calls | total us | avg us | min us | max us | share |")
        print("|---|---|---|---|---|---|---|")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"| `{k}` | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f}% |")
        print()
    if counters:
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in counters:
            for r in csv.DictReader(open(f)):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        print("## counters (average per dispatch)\n")
        print("| kernel | counter | dispatches | avg value | sum |")
        print("|---|---|---|---|---|")
        for k in sorted(agg):
            for c, a in sorted(agg[k].items()):
                print(f"| `{k}` | {c} | {a[0]} | {a[1] / a[0]:.4g} | {a[1]:.6g} |")


def traffic(fetch_dir, write_dir, key, out_json, source=""):
    """HBM bytes per conv3x3 launch = 2 * FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KB per dispatch, from two separate --pmc passes."""
    import json

    def total(d, counter):
        n, v = 0, 0.0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv3x3_" in r["Kernel_Name"] and "pack_" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    n += 1
                    v += float(r["Counter_Value"])
        return n, v

    nf, f = total(fetch_dir, "FETCH_SIZE")
    nw, w = total(write_dir, "WRITE_SIZE")
    per_launch = (2.0 * f / nf + w / nw) * 1024.0
    try:
        d = json.load(open(out_json))
    except Exception:
        d = {}
    d[key] = {"hbm_bytes_per_conv3x3_launch": round(per_launch), "fetch_kb_per_launch_raw": round(f / nf, 1),
              "write_kb_per_launch": round(w / nw, 1), "dispatches": nf,
              "source": source,
              "note": "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads), two separate "
                      "rocprofv3 --pmc passes of the bench command, averaged over every conv3x3_* dispatch; FETCH_SIZE counts L2 misses, "
                      "including those the Infinity Cache serves"}
    json.dump(d, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps(d[key]))


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        traffic(*sys.argv[2:7])
    else:
        main(sys.argv[1])
