#!/bin/bash
# Same-box A/B of the product library against a previous build kept as applied-image-processing_amd/libadain_hip_prev.so (untracked):
# parity tests of the new build first, then bench.py of both, interleaved, two runs per config.  Usage: bash tools/probes/ab_prev.sh [configs...]
set -e
P=applied-image-processing_amd/libadain_hip_prev.so
cfgs="${*:-2 4 5 3}"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/ab_parity.log 2>&1 || { tail -20 gpurun_out/ab_parity.log; exit 1; }
tail -2 gpurun_out/ab_parity.log
for rep in 1 2; do
  for cfg in $cfgs; do
    python bench.py --config $cfg --no-cpu --no-secondary --layers --lib $P > gpurun_out/ab_prev_c${cfg}_$rep.json 2> gpurun_out/ab_prev_c${cfg}_$rep.err
    python bench.py --config $cfg --no-cpu --no-secondary --layers > gpurun_out/ab_new_c${cfg}_$rep.json 2> gpurun_out/ab_new_c${cfg}_$rep.err
  done
done
python - "$cfgs" <<'PY'
import json, glob, sys
for cfg in sys.argv[1].split():
    for w in ("prev", "new"):
        v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/ab_{w}_c{cfg}_*.json"))]
        print(cfg, w, [(d["value"], d["ms_per_step"], d["roofline"]["frac"]) for d in v])
PY
