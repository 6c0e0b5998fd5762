set -e
P=applied-image-processing_amd/libadain_hip_prev.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_ab_parity.log 2>&1 || { tail -20 gpurun_out/r3_ab_parity.log; exit 1; }
tail -2 gpurun_out/r3_ab_parity.log
for rep in 1 2; do
  for cfg in 2 4 5 3; do
    python bench.py --config $cfg --no-cpu --no-secondary --layers --lib $P > gpurun_out/r3_ab_prev_c${cfg}_$rep.json 2> gpurun_out/r3_ab_prev_c${cfg}_$rep.err
    python bench.py --config $cfg --no-cpu --no-secondary --layers > gpurun_out/r3_ab_new_c${cfg}_$rep.json 2> gpurun_out/r3_ab_new_c${cfg}_$rep.err
  done
done
python - <<'PY'
import json,glob
for cfg in (2,4,5,3):
    for w in ("prev","new"):
        v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r3_ab_{w}_c{cfg}_*.json"))]
        print(cfg,w,[ (d["value"],d["ms_per_step"],d["roofline"]["frac"]) for d in v])
PY
