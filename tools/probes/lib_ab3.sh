#!/bin/bash
# Same-box A/B/C of builds of the product library through bench.py --lib: tools/probes/lib_ab3.sh "<lib1> <lib2> ..." [bench args...]
cd "${GRAFT_REPO_ROOT:-$PWD}"
libs="$1"; shift
for round in 1 2; do
  for lib in product $libs; do
    arg=""; [ "$lib" != product ] && arg="--lib $lib"
    out=$(python bench.py $arg --no-cpu --no-secondary --sustain 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], [(x['kernel'][:12], x['avg_us']) for x in d['secondary'][:2]])")
    echo "$lib: $out"
  done
done
