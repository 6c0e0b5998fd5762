#!/bin/bash
# What the halo staging costs the config-2 step: timing-only builds of the PERSISTENT kernel inside the whole pass (diagnostic library,
# ADAIN_W4_PDIAG; results are wrong by construction).  0 = product code, 11 = no halo stores, 10 = no halo loads, 8 = neither.
for d in 0 11 10 8 0; do
  ADAIN_W4_PDIAG=$d python bench.py --diag-lib --no-cpu --no-secondary --steps 20 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('PDIAG=$d', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])"
done
