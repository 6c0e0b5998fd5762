#!/usr/bin/env python3
"""Turns one collection run merged back under gpurun_out/ into the committed evidence under profiles/ for a round:

    python tools/refresh_profiles.py <round, e.g. r03> <bench tag> <profile tag>

bench lines   gpurun_out/<bench tag>_{c2,c2_pcie,c3,c4,c5,c4job,c5job,c2_ws1,c4job_ws1,c5job_ws1}.log   (tools/gpu_steps.sh)
profiles      gpurun_out/prof_<profile tag>_*                                             (tools/collect_profiles.sh, collect_traffic.sh)
-> profiles/<round>_bench_*.json, <round>_kernel_trace*.md, <round>_kernel_stats.csv, <round>_all_kernels.md, <round>_pmc.md,
   <round>_gaps.md, pmc_traffic.json.  Every file's header names the commit it was taken at (HEAD when this script runs: run it
   before changing the code again)."""
import csv
import glob
import io
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gap_report  # noqa: E402
import summarize_rocprof as sr  # noqa: E402

rnd, btag, ptag = sys.argv[1:4]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
# the commit the profiles name = the code commit of the manifest they were collected under; refused if the product sources have moved since
_st = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_manifest.py"), "--stamp"], capture_output=True, text=True)
if _st.returncode != 0:
    sys.exit(_st.stderr.strip() or _st.stdout.strip())
HEAD = _st.stdout.strip().split("\n")[-1]
STAMP = f"(commit {HEAD}; 1 x MI355X box of the gpurun pool, ROCm 7.2, rocprofv3)"


def last_json(name):
    return [l for l in open(os.path.join(G, name)).read().strip().split("\n") if l.startswith("{")][-1]


def captured(fn, *a):
    buf = io.StringIO()
    with redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def counters(d):
    out = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = out[sr.short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return out


# ---- bench lines ------------------------------------------------------------------------------------------------------------------
lines = {"c2": "config2", "c2_pcie": "config2_pcie", "c3": "config3", "c4": "config4", "c5": "config5", "c4job": "config4_job",
         "c5job": "config5_job", "c2_ws1": "config2_ws1_rccl", "c4job_ws1": "config4_job_ws1_rccl", "c5job_ws1": "config5_job_ws1_rccl",
         "c2_ws1_overlap": "config2_ws1_rccl_gather_overlap", "c2_style1024": "config2_style1024", "c4_depth_job": "config4_depth_job",
         "c2_tl": "config2_trained_like", "c4job_tl": "config4_job_trained_like", "pc_video": "per_call_video", "pc_video_lat": "per_call_video_latency", "pc_guide": "per_call_guide", "reh2_end": "rehearsal_2ranks_one_gpu", "reh3_job": "rehearsal_3ranks_one_gpu_job5_chunked"}
bench = {}
for src, dst in lines.items():
    path = os.path.join(G, f"{btag}_{src}.log")
    if os.path.exists(path):
        bench[src] = json.loads(last_json(f"{btag}_{src}.log"))
        open(os.path.join(P, f"{rnd}_bench_{dst}.json"), "w").write(json.dumps(bench[src]) + "\n")

# ---- kernel traces -------------------------------------------------------------------------------------------------------------------
stats = glob.glob(os.path.join(G, f"prof_{ptag}_trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(P, f"{rnd}_kernel_stats.csv"))
b = json.loads(last_json(f"prof_{ptag}_trace.log"))
tot = n = 0
for f in glob.glob(os.path.join(G, f"prof_{ptag}_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3_wino4" in r["Kernel_Name"]:
            tot += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n += 1
u = bench.get("c2", b)
open(os.path.join(P, f"{rnd}_kernel_trace.md"), "w").write(
    f"# {rnd} — rocprofv3 kernel trace of `bench.py --no-cpu --steps 10 --warmup 2` (config 2, with the 1080p pixel-kernel leg) {STAMP}\n\n"
    "`tools/collect_profiles.sh <tag> trace`: `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d … -- python3 bench.py --no-cpu "
    "--steps 10 --warmup 2` (2 warm-up + 10 timed steps, the `sustained` leg - the same loop for two more seconds -, 7 event-instrumented steps, then the frame-sized pixel kernels of the video / guide\n"
    "post-pass on 8 x 1080p: `resize_area*`, `warp_blend_u8*`, `quantize_u8*`, `mask_composite`, `resize_*`).\n"
    f"bench line of the same (profiled) run: {b['value']:.2f} Mpixels/s, {b['ms_per_step']:.3f} ms/step, roofline.avg_launch_ms "
    f"{b['roofline']['avg_launch_ms']:.4f} (HIP events); unprofiled run of the same binary: profiles/{rnd}_bench_config2.json "
    f"({u['value']:.2f} Mpixels/s, {u['ms_per_step']:.3f} ms/step, avg_launch_ms {u['roofline']['avg_launch_ms']:.4f}).\n"
    f"conv3x3_wino4_kernel over all dispatches: {tot:,.1f} us / {n} dispatches = {tot / n:.1f} us per launch under the profiler "
    "(16 launches per step: 8 merged content+style encoder layers, 8 decoder layers).\n\n" + captured(sr.main, os.path.join(G, f"prof_{ptag}_trace")))
open(os.path.join(P, f"{rnd}_gaps.md"), "w").write(
    f"# {rnd} — idle time between the kernels of the config-2 step {STAMP}\n\nFrom the same trace as {rnd}_kernel_trace.md (`tools/gap_report.py`, "
    "the first 100 dispatches = weight packing and warm-up skipped): start of the next kernel minus end of the previous one.\n\n"
    + captured(gap_report.main, os.path.join(G, f"prof_{ptag}_trace"), 100))
for cfg in (3, 4, 5):
    d = f"prof_{ptag}_cfg{cfg}_trace"
    if os.path.isdir(os.path.join(G, d)):
        bl = json.loads(last_json(d + ".log"))
        open(os.path.join(P, f"{rnd}_kernel_trace_config{cfg}.md"), "w").write(
            f"# {rnd} — rocprofv3 kernel trace of `bench.py --config {cfg} --no-cpu --no-secondary --steps 5 --warmup 1` {STAMP}\n\n"
            f"bench line of the profiled run: {bl['value']:.2f} Mpixels/s, {bl['ms_per_step']:.3f} ms/step, matrix pipe {bl['roofline']['frac']:.4f}; "
            f"`<…, 1>` = the 16 x 16 tile geometry, `<…, 0>` = 8 x 32 (csrc/conv_wino4.hip).\n\n" + captured(sr.main, os.path.join(G, d)))
d = f"prof_{ptag}_all_kernels"
if os.path.isdir(os.path.join(G, d)):
    open(os.path.join(P, f"{rnd}_all_kernels.md"), "w").write(
        f"# {rnd} — every kernel of the product library in one trace: `tools/all_kernels.py` {STAMP}\n\n"
        "`cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d … -- python3 tools/all_kernels.py`: every public entry point at shapes\n"
        "that select each launch form (both tile geometries, persistent / one-tile, up-sampling, uint8 / float first layer, NHWC / NCHW statistics,\n"
        "every INTER_AREA form ...).  Not in this trace: the `BIG` instantiations of conv3x3_wino4_kernel (per-image tensors >= 2 GiB; exercised by\n"
        "tests/test_gpu_parity.py::test_conv_tensors_above_two_gib).  `__amd_rocclr_*` / `at::native::*` rows are the script's own tensor set-up.\n\n"
        + open(os.path.join(G, d + ".log")).read().strip().split("\n")[-1] + "\n\n" + captured(sr.main, os.path.join(G, d)))

# ---- counters ----------------------------------------------------------------------------------------------------------------------
pmc = [f"# {rnd} — rocprofv3 PMC passes, one counter set per run, config 2 unless stated {STAMP}\n",
       "`tools/collect_profiles.sh <tag> pmc waits`; `bench.py --no-cpu --no-secondary --steps 5 --warmup 1` under `rocprofv3 --pmc <counters> "
       "--kernel-trace` (never combined with other trace domains).  FETCH_SIZE / WRITE_SIZE in KB per dispatch; 2 x FETCH_SIZE + WRITE_SIZE = "
       "L2-miss traffic (gfx950 reports half the bytes of wide coalesced reads; it includes what the 256 MB Infinity Cache serves).\n"]
for part in ("fetch", "write", "sq"):
    d = os.path.join(G, f"prof_{ptag}_{part}")
    if os.path.isdir(d):
        s = captured(sr.main, d)
        pmc.append(f"\n## pass `{part}`\n\n" + s[s.index("## counters") + len("## counters (average per dispatch)\n"):])
sq = counters(f"prof_{ptag}_sq")
rows = []
for k, c in sorted(sq.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and "conv3x3_wino4" in k:
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (c["GRBM_GUI_ACTIVE"][1] / 8 * 1024)     # per-SIMD busy cycles / (chip cycles x 1024 SIMDs)
        vm = c["SQ_INSTS_VALU"][1] / c["SQ_INSTS_MFMA"][1]
        lds = c["SQ_LDS_BANK_CONFLICT"][1] / max(c["SQ_LDS_IDX_ACTIVE"][1], 1)
        rows.append(f"| `{k}` | {c['SQ_INSTS_MFMA'][0]} | {busy:.4f} | {vm:.2f} | {c['SQ_INSTS_MFMA'][1] / c['SQ_INSTS_MFMA'][0] * 4096 / 1e9:.3f} | {100 * lds:.1f} % |")
if rows:
    pmc.append("\n## derived (pass `sq`)\n\nmatrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); executed GFLOP = SQ_INSTS_MFMA x 4096 flop\n\n"
               "| kernel | dispatches | matrix pipe busy | SQ_INSTS_VALU / SQ_INSTS_MFMA | executed GFLOP per dispatch | LDS bank-conflict cycles |\n|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
w = {cfg: (counters(f"prof_{ptag}_cfg{cfg}_waits"), counters(f"prof_{ptag}_cfg{cfg}_tcc")) for cfg in (2, 3)}
rows = []
for cfg, (wa, tc) in w.items():
    for k in sorted(wa):
        if "conv3x3_wino4_kernel<0, 0, true" in k and k in tc:
            a, t = wa[k], tc[k]
            rows.append(f"| {cfg} | `{k}` | {a['SQ_WAVE_CYCLES'][0]} | {a['SQ_WAIT_ANY'][1] / a['SQ_WAVE_CYCLES'][1]:.4f} | "
                        f"{a['SQ_WAIT_INST_ANY'][1] / a['SQ_WAVE_CYCLES'][1]:.4f} | {t['TCC_HIT_sum'][1] / (t['TCC_HIT_sum'][1] + t['TCC_MISS_sum'][1]):.4f} | "
                        f"{t['TCC_MISS_sum'][1] / t['TCC_MISS_sum'][0]:.3g} |")
if rows:
    pmc.append("\n## is the larger feature map's extra L2-miss traffic free?  (passes `cfg{2,3}_waits`, `cfg{2,3}_tcc`; direct persistent launches)\n\n"
               "Config 3 (2048 x 2048) misses the 4 MB L2s more often than config 2 (the 256-channel layers work on 512 x 512 maps: a round of 64 resident "
               "workgroups per XCD streams 5.6 MB of halos beside its 3.1 MB of weights), but its waves spend no larger share of their cycles waiting: the "
               "misses are served by the Infinity Cache behind the second wave of each SIMD, and the same-box bench lines run config 3 at a matrix-pipe "
               "fraction no lower than config 2's.\n\n"
               "| config | kernel | dispatches | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | L2 hit rate TCC_HIT / (HIT + MISS) | TCC_MISS per dispatch |\n"
               "|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
open(os.path.join(P, f"{rnd}_pmc.md"), "w").write("\n".join(pmc))

# ---- HBM-side traffic per conv3x3 launch (roofline.traffic of the bench lines) ---------------------------------------------------------
tj = os.path.join(P, "pmc_traffic.json")
src = f"profiles/{rnd}_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py {{args}}--no-cpu --no-secondary --steps 5 --warmup 1, commit {HEAD})"
for key, f, wdir, args in (("config2_batch1", f"prof_{ptag}_fetch", f"prof_{ptag}_write", ""),
                           ("config3_batch1", f"prof_{ptag}_cfg3_fetch", f"prof_{ptag}_cfg3_write", "--config 3 "),
                           ("config4_batch1", f"prof_{ptag}_cfg4_fetch", f"prof_{ptag}_cfg4_write", "--config 4 "),
                           ("config5_batch1", f"prof_{ptag}_cfg5_fetch", f"prof_{ptag}_cfg5_write", "--config 5 "),
                           ("config4_batch2", f"prof_{ptag}_cfg4b2_fetch", f"prof_{ptag}_cfg4b2_write", "--config 4 --batch 2 "),
                           ("config5_batch2", f"prof_{ptag}_cfg5b2_fetch", f"prof_{ptag}_cfg5b2_write", "--config 5 --batch 2 ")):
    if os.path.isdir(os.path.join(G, f)) and os.path.isdir(os.path.join(G, wdir)):
        captured(sr.traffic, os.path.join(G, f), os.path.join(G, wdir), key, tj, src.format(args=args))
print("profiles refreshed for", rnd, "at", HEAD)
for f in sorted(os.listdir(P)):
    if f.startswith(rnd):
        print("  ", f, os.path.getsize(os.path.join(P, f)))
