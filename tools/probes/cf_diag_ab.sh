#!/bin/bash
# Round 5: timing-only ablations of conv_first_kernel in the diagnostic library (ADAIN_CF_DIAG: 1 no global stores, 2 no MFMAs, 3 neither)
# and the workgroups-per-CU switch (ADAIN_CF_WGS), read from bench.py's `secondary` table (HIP events, 1024 x 1024, float entry).
cd "${GRAFT_REPO_ROOT:-$PWD}"
for cfd in 0 1 2 3; do
  for wgs in ${CF_WGS_LIST:-3}; do
    out=$(ADAIN_CF_DIAG=$cfd ADAIN_CF_WGS=$wgs python bench.py --diag-lib --no-cpu --no-secondary --sustain 0 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print([ (x['kernel'][:18], x['avg_us']) for x in d['secondary'] if 'conv_first' in x['kernel'] or 'conv_last' in x['kernel']])")
    echo "ADAIN_CF_DIAG=$cfd ADAIN_CF_WGS=$wgs: $out"
  done
done
