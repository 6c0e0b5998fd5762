#!/bin/bash
# conv_first_kernel in the diagnostic library, read from bench.py's `secondary` table (HIP events, 1024 x 1024, float entry):
#   ADAIN_CF_DIAG   timing-only ablations (round 5): 1 no global stores, 2 no MFMAs, 3 neither
#   ADAIN_CF_DEEP   the two-tiles-ahead halo prefetch (round 6), A B A B against the product form, then the uint8 entry at 1080p
#   ADAIN_CF_WGS    workgroups per CU (CF_WGS_LIST, default 3)
# The switches are compiled into the diagnostic library (csrc/conv_edge.hip, -DADAIN_DIAG); the script FAILS if an ablation does not
# change the time (round-5 advisor finding: a script that measures the same kernel four times must say so).
cd "${GRAFT_REPO_ROOT:-$PWD}"
read_us() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(' '.join(str(x['avg_us']) for x in d['secondary'] if 'conv_first' in x['kernel']))"; }
declare -A us
for cfd in 0 1 2 3; do
  for wgs in ${CF_WGS_LIST:-3}; do
    out=$(ADAIN_CF_DIAG=$cfd ADAIN_CF_WGS=$wgs python bench.py --diag-lib --no-cpu --no-secondary --sustain 0 --steps 10 2>/dev/null | read_us)
    echo "ADAIN_CF_DIAG=$cfd ADAIN_CF_WGS=$wgs: conv_first $out us"
    us[$cfd]=$out
  done
done
python - "${us[0]}" "${us[3]}" <<'PY' || exit 1
import sys
whole, neither = float(sys.argv[1].split()[0]), float(sys.argv[2].split()[0])
if not neither < 0.7 * whole:
    print(f"the ablation switch has no effect ({whole} vs {neither} us): is this the diagnostic library of this tree?"); sys.exit(1)
PY
for round in 1 2; do
  for deep in 0 1; do
    a=$(ADAIN_CF_DEEP=$deep python bench.py --diag-lib --no-cpu --no-secondary --sustain 0 --steps 20 2>/dev/null | read_us)
    b=$(ADAIN_CF_DEEP=$deep python bench.py --diag-lib --config 4 --no-cpu --no-secondary --sustain 0 --steps 20 2>/dev/null | read_us)
    echo "ADAIN_CF_DEEP=$deep: 1024 x 1024 float entry $a us | 1080p uint8 entry $b us"
  done
done
a=$(ADAIN_CF_DEEP=1 ADAIN_CF_DIAG=3 python bench.py --diag-lib --no-cpu --no-secondary --sustain 0 --steps 10 2>/dev/null | read_us)
echo "ADAIN_CF_DEEP=1 ADAIN_CF_DIAG=3 (neither stores nor MFMAs, loads two tiles ahead): $a us"
