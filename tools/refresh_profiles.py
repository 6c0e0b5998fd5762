#!/usr/bin/env python3
"""Turns one collection run (tools/gpu_steps.sh bench lines `<tag>_c*.log` + tools/collect_profiles.sh `prof_<tag>_*`, merged
back under gpurun_out/) into the committed evidence under profiles/: r02_bench_*.json, r02_kernel_trace.md, r02_kernel_stats.csv,
r02_pmc.md (the appended probe sections are kept), pmc_traffic.json.   python tools/refresh_profiles.py <tag>"""
import csv
import glob
import io
import json
import os
import shutil
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_rocprof as sr  # noqa: E402

tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def last_json(name):
    return [l for l in open(os.path.join(G, name)).read().strip().split("\n") if l.startswith("{")][-1]


def summary(d):
    buf = io.StringIO()
    with redirect_stdout(buf):
        sr.main(os.path.join(G, d))
    return buf.getvalue()


def counters_of(d):
    s = summary(d)
    return s[s.index("## counters"):]


for src, dst in (("c2", "config2"), ("c3", "config3"), ("c4", "config4"), ("c5", "config5"), ("c2_ws1", "config2_ws1_rccl"),
                 ("c4_ws1", "config4_ws1_rccl"), ("c2_pcie", "config2_pcie")):
    open(os.path.join(P, f"r02_bench_{dst}.json"), "w").write(last_json(f"{tag}_{src}.log") + "\n")
stats = glob.glob(os.path.join(G, f"prof_{tag}_trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(P, "r02_kernel_stats.csv"))
b, u = json.loads(last_json(f"prof_{tag}_trace.log")), json.loads(last_json(f"{tag}_c2.log"))
tot = n = 0
for f in glob.glob(os.path.join(G, f"prof_{tag}_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3_wino4" in r["Kernel_Name"]:
            tot += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n += 1
trace = summary(f"prof_{tag}_trace")
open(os.path.join(P, "r02_kernel_trace.md"), "w").write(
    "# Round 2 — rocprofv3 kernel trace, bench.py --no-cpu --no-secondary --steps 10 --warmup 2 (1 MI355X, config 2, final round-2 binary)\n\n"
    "`tools/collect_profiles.sh`: `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d … -- python3 bench.py --no-cpu "
    "--no-secondary --steps 10 --warmup 2` (19 steps traced: 2 warm-up + 10 timed + 7 event-instrumented).\n"
    f"bench line of the same (profiled) run: {b['value']:.2f} Mpixels/s, {b['ms_per_step']:.3f} ms/step, roofline.avg_launch_ms "
    f"{b['roofline']['avg_launch_ms']:.4f} (HIP events); unprofiled run of the same binary: profiles/r02_bench_config2.json "
    f"({u['value']:.2f} Mpixels/s, {u['ms_per_step']:.3f} ms/step, avg_launch_ms {u['roofline']['avg_launch_ms']:.4f}).\n"
    f"conv3x3_wino4_kernel over all dispatches: {tot:,.1f} us / {n} dispatches = {tot / n:.1f} us per launch under the profiler "
    "(16 launches per step: 8 merged content+style encoder layers, 8 decoder layers).\n\n" + trace)


def last_step(d, counter):
    rows = [r for f in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True) for r in csv.DictReader(open(f))]
    rows = [r for r in rows if r["Counter_Name"] == counter and "conv3x3_wino4" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows[-16:]


names = ["conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "conv4_1", "dec1", "dec2", "dec3", "dec4", "dec5",
         "dec6", "dec7", "dec8"]
tab = ("## per conv3x3 launch of the last step (MB: 2 x FETCH_SIZE, WRITE_SIZE); encoder launches carry content + style\n\n"
       "| layer | fetch MB | write MB |\n|---|---|---|\n")
for nm, a, w in zip(names, last_step(f"prof_{tag}_fetch", "FETCH_SIZE"), last_step(f"prof_{tag}_write", "WRITE_SIZE")):
    tab += f"| {nm} | {2 * float(a['Counter_Value']) / 1024:.1f} | {float(w['Counter_Value']) / 1024:.1f} |\n"

# ratios quoted in the header, from the sq pass
agg = {}
for f in glob.glob(os.path.join(G, f"prof_{tag}_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = sr.short(r["Kernel_Name"])
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        agg[k][r["Counter_Name"]][0] += 1
        agg[k][r["Counter_Name"]][1] += float(r["Counter_Value"])


def ratio(k, a, bb, scale=1.0):
    return agg[k][a][1] / (agg[k][bb][1] * scale)


ks = ["conv3x3_wino4_kernel<0, 0, true, false>", "conv3x3_wino4_kernel<1, 0, true, false>", "conv3x3_wino4_kernel<0, 0, false, false>"]
ks = [k if k in agg else k.replace(", false>", ">") for k in ks]
busy = [ratio(k, "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", 128.0) for k in ks]
vpm = [ratio(k, "SQ_INSTS_VALU", "SQ_INSTS_MFMA") for k in ks]
lds = [ratio(k, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") for k in ks]
edge = {k: ratio(k, "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", 128.0) for k in agg if k.startswith("conv_first") or k.startswith("conv_last")}
fetch_last = [r for f in glob.glob(os.path.join(G, f"prof_{tag}_fetch", "**", "*counter_collection.csv"), recursive=True)
              for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "conv_last_kernel" in r["Kernel_Name"]]
cl_mb = 2 * sum(float(r["Counter_Value"]) for r in fetch_last) / len(fetch_last) / 1024
old = subprocess.run(["git", "show", "HEAD:profiles/r02_pmc.md"], capture_output=True, text=True, cwd=ROOT).stdout
keep = old[old.index("## 256->256 @256^2 probe"):old.index("## per conv3x3 launch of the last step")]
hdr = ("# Round 2 — rocprofv3 PMC passes (separate runs) of bench.py --no-cpu --no-secondary --steps 5 --warmup 1 (config 2, final round-2 binary)\n\n"
       "`tools/collect_profiles.sh` (each pass: `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py …`, nothing else traced); summarised by `tools/refresh_profiles.py`.\n"
       "Units: FETCH_SIZE / WRITE_SIZE in KB per dispatch (FETCH_SIZE is doubled in profiles/pmc_traffic.json: gfx950 counts half the bytes of wide "
       "coalesced reads; it counts every L2 miss, including those the Infinity Cache serves).  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 "
       f"SIMDs) = matrix-pipe busy fraction: {busy[0]:.3f} (persistent, direct source), {busy[1]:.3f} (persistent, fused 2x upsample), {busy[2]:.3f} "
       "(one-tile form: dec1); " + ", ".join(f"{k.split('<')[0]} {v:.2f}" for k, v in sorted(edge.items())) + ".  SQ_INSTS_VALU / SQ_INSTS_MFMA = "
       f"{vpm[0]:.1f} / {vpm[1]:.1f} / {vpm[2]:.1f}: {vpm[0] - 1:.1f} / {vpm[1] - 1:.1f} / {vpm[2] - 1:.1f} plain vector instructions per MFMA over whole "
       "launches (3.2 inside the main loop; the rest is the per-tile epilogue and prologue, which weighs most on the cin = 64 layers).  "
       f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {lds[0]:.2f} / {lds[1]:.2f} / {lds[2]:.2f} with the de-interleaved halo image (the interleaved image "
       f"of round 1: 0.49 on the 256->256 probe; same probe now 0.08).  conv_last: 2 x FETCH_SIZE = {cl_mb:.0f} MB per launch for 268 MB of input "
       "(XCD-contiguous tile ranges; the round-1 kernel fetched 323 MB).\n\n")
body = ("## pass 1: FETCH_SIZE\n" + counters_of(f"prof_{tag}_fetch") + "\n## pass 2: WRITE_SIZE\n" + counters_of(f"prof_{tag}_write")
        + "\n## pass 3: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA\n"
        + counters_of(f"prof_{tag}_sq"))
open(os.path.join(P, "r02_pmc.md"), "w").write(hdr + body.rstrip("\n") + "\n\n" + keep + tab)
sr.traffic(os.path.join(G, f"prof_{tag}_fetch"), os.path.join(G, f"prof_{tag}_write"), "config2_batch1", os.path.join(P, "pmc_traffic.json"),
           "profiles/r02_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --no-cpu --no-secondary --steps 5 --warmup 1, round 2 final binary)")
print("busy", [round(x, 3) for x in busy], "valu/mfma", [round(x, 2) for x in vpm], "lds", [round(x, 3) for x in lds], "edge", edge)
print("trace:", f"{tot:,.1f} us / {n} = {tot / n:.1f} us; profiled events {b['roofline']['avg_launch_ms']:.4f}")
