#!/usr/bin/env python3
"""Inter-kernel gaps of a rocprofv3 kernel trace: for every dispatch of the traced run, its duration and the idle time
before the next kernel starts (start[i+1] - end[i]); a table per kernel name plus the share of the wall time between the first
start and the last end of the steady-state part that no kernel covers.

    python tools/gap_report.py <rocprof output dir> [skip_first_n_dispatches] > profiles/<name>.md
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("adain::", "")[:70]


def main(d, skip=0):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[skip:]
    # the steps only: from the first to the last generic 3x3 convolution (+ the last layer behind it); whatever the traced command
    # runs before (weight packing, warm-up) or after (bench.py's pixel-kernel leg) is not part of a step
    conv = [i for i, r in enumerate(rows) if "conv3x3_" in r[2]]
    if conv:
        rows = rows[conv[0]:min(conv[-1] + 2, len(rows))]
    agg = defaultdict(lambda: [0, 0.0, []])
    busy = 0.0
    for i, (s, e, k) in enumerate(rows):
        gap = (rows[i + 1][0] - e) / 1e3 if i + 1 < len(rows) else 0.0
        a = agg[k]
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2].append(max(gap, 0.0))
        busy += (e - s) / 1e3
    wall = (rows[-1][1] - rows[0][0]) / 1e3
    print(f"dispatches {len(rows)}, wall {wall:.1f} us, kernels {busy:.1f} us ({100 * busy / wall:.1f} %)\n")
    print("The traced command also runs event-instrumented steps (a hipEventRecord between kernels: 2-4 us each) and synchronises the host between\n"
          "them, so MEANS are dominated by those; the MEDIAN is the back-to-back case of the timed loop.\n")
    print("| kernel | calls | avg us | median gap to the next kernel us | 90th percentile us |")
    print("|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        g = sorted(a[2])
        print(f"| `{k}` | {a[0]} | {a[1] / a[0]:.2f} | {g[len(g) // 2]:.2f} | {g[min(len(g) - 1, int(0.9 * len(g)))]:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
