#!/bin/bash
# Same-box A/B of two builds of the product library through bench.py --lib: tools/probes/lib_ab.sh <other.so> [bench args...]
# prints value, ms/step and the secondary table's conv_first / conv_last / mean_std rows for each, A B A B.
cd "${GRAFT_REPO_ROOT:-$PWD}"
other="$1"; shift
for round in 1 2; do
  for lib in "" "--lib $other"; do
    out=$(python bench.py $lib --no-cpu --no-secondary --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], [(x['kernel'][:12], x['avg_us']) for x in d['secondary'][:4]])")
    echo "${lib:-product}: $out"
  done
done
