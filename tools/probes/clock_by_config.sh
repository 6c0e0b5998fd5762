#!/bin/bash
# Round 5: which shader clock does the chip hold under each workload's launch pattern?  rocm-smi sampled every 100 ms beside the
# `sustained` leg of bench.py (6 s of back-to-back steps).   tools/probes/clock_by_config.sh "2:1" "4:1" "3:1" "4:4" ...  (config:batch)
cd "${GRAFT_REPO_ROOT:-$PWD}"
for item in "$@"; do
  IFS=: read cfg batch <<< "$item"
  log=gpurun_out/clock_cfg${cfg}_b${batch}_$RANDOM.smi
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 >> "$log"; sleep 0.1; done ) &
  sampler=$!
  out=$(python bench.py --config $cfg --batch $batch --no-cpu --no-secondary --sustain 6 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['sustained']['value'], d['roofline']['frac'])")
  kill $sampler 2>/dev/null; wait $sampler 2>/dev/null
  python - "$log" "$cfg:$batch" "$out" <<'PY'
import re, sys
log, item, out = sys.argv[1:4]
sclk, power = [], []
for ln in open(log):
    m = re.findall(r"\((\d+)Mhz\)", ln)
    nums = re.findall(r"(?<![\w.])(\d+\.\d+)(?![\w.])", ln)
    if len(m) >= 3:
        sclk.append(int(m[2]))
    if nums:
        power.append(float(nums[-1]))
busy = [c for c, p in zip(sclk, power) if p > 1000]
bp = [p for p in power if p > 1000]
print(f"config:batch {item}: value, ms/step, sustained, frac = {out} | loaded samples {len(busy)}: sclk median {sorted(busy)[len(busy)//2] if busy else None} MHz mean {sum(busy)/max(len(busy),1):.0f} | power median {sorted(bp)[len(bp)//2] if bp else None} W")
PY
done
