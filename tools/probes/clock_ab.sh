#!/bin/bash
# Round 5: does the batch schedule change the CLOCK the chip holds?  Same-box A/B of the two batch schedules of csrc/api.hip in the
# diagnostic library (frame-major big layers / every layer over the whole batch) on 1080p batches, with rocm-smi sampled every 100 ms
# beside each run (sclk, power): if the layer-major batch runs at a lower clock, the few per cent it loses on the mid-network layers is
# power management, not the memory system.   tools/probes/clock_ab.sh [batch] [steps]
cd "${GRAFT_REPO_ROOT:-$PWD}"
batch=${1:-4}; steps=${2:-300}
for rounds in 60 10000000 60 10000000; do
  log=gpurun_out/clock_ab_${rounds}_$RANDOM.smi
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 >> "$log"; sleep 0.1; done ) &
  sampler=$!
  out=$(ADAIN_BIG_FRAME_WIDTH=0 ADAIN_BIG_ROUNDS_X10=$rounds python bench.py --diag-lib --config 4 --batch $batch --no-cpu --no-secondary --sustain 0 --steps $steps --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])")
  kill $sampler 2>/dev/null; wait $sampler 2>/dev/null
  python - "$log" "$rounds" "$out" <<'PY'
import re, sys
log, rounds, out = sys.argv[1:4]
sclk, power = [], []
for ln in open(log):
    m = re.findall(r"\((\d+)Mhz\)", ln)
    nums = re.findall(r"(?<![\w.])(\d+\.\d+)(?![\w.])", ln)
    if m:
        sclk.append(int(m[-1]) if len(m) == 1 else max(int(v) for v in m))
    if nums:
        power.append(float(nums[-1]))
busy = sorted(sclk)[len(sclk) // 4:] if sclk else []
print(f"rounds_x10 {rounds:>8}: {out} Mpixels/s, ms per step | sclk samples {len(sclk)}: median {sorted(sclk)[len(sclk)//2] if sclk else None} MHz, upper-3/4 mean {sum(busy)/max(len(busy),1):.0f} | power median {sorted(power)[len(power)//2] if power else None} W")
PY
  head -3 "$log" | cut -c1-200
done
