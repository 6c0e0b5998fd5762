#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, nothing else traced) of bench.py for one config -> gpurun_out/prof_<tag>_cfg<C>[b<B>]_{fetch,write}
# Usage: tools/collect_traffic.sh <tag> <config>[:<batch>] [...]        e.g. 3 4 5 4:2 5:2
set -o pipefail
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out"
mkdir -p "$out"
python3 "$root/tools/profile_manifest.py" --check || exit 2
cd /tmp; export TMPDIR=/tmp
for item in "$@"; do
  cfg="${item%%:*}"; batch=1; [ "$item" != "$cfg" ] && batch="${item#*:}"
  suffix=""; [ "$batch" != 1 ] && suffix="b$batch"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    name="cfg${cfg}${suffix}_$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )"
    echo "=== $name"
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/prof_${tag}_$name" -- python3 "$root/bench.py" --config "$cfg" \
        --batch "$batch" --no-cpu --no-secondary --steps 5 --warmup 1 --sustain 2 > "$out/prof_${tag}_$name.log" 2> "$out/prof_${tag}_$name.err"
    rc=$?; echo "=== $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
find "$out" -path "*prof_${tag}_cfg*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.err" -delete
