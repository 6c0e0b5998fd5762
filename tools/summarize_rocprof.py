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


if __name__ == "__main__":
    main(sys.argv[1])
