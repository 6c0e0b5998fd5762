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
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    busy = 0.0
    for i, (s, e, k) in enumerate(rows):
        gap = (rows[i + 1][0] - e) / 1e3 if i + 1 < len(rows) else 0.0
        a = agg[k]
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] += max(gap, 0.0) if gap < 200 else 0.0       # gaps above 200 us are host pauses (between steps of the CPU leg), not dispatch gaps
        busy += (e - s) / 1e3
    wall = (rows[-1][1] - rows[0][0]) / 1e3
    print(f"dispatches {len(rows)}, wall {wall:.1f} us, kernels {busy:.1f} us ({100 * busy / wall:.1f} %), "
          f"dispatch gaps {sum(a[2] for a in agg.values()):.1f} us\n")
    print("| kernel | calls | avg us | avg gap to the next kernel us |")
    print("|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / a[0]:.2f} | {a[2] / a[0]:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
