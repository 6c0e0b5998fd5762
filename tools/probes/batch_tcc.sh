#!/bin/bash
# L2 hit / miss counters per conv dispatch for 1080p frames at batch 1 and batch 2 with every layer run over the whole batch (the
# layer-major schedule, diagnostic library: ADAIN_BIG_ROUNDS_X10=10000000): does a batch miss the L2 more often per frame?
root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
export ADAIN_BIG_ROUNDS_X10=10000000 ADAIN_BIG_FRAME_WIDTH=0
for b in 1 2; do
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    name="tcc_b${b}_$(echo $ctr | cut -c1-5)"
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/prof_$name" -- python3 "$root/bench.py" --diag-lib --config 4 --batch $b \
        --no-cpu --no-secondary --steps 3 --warmup 1 --sustain 0 > "$out/prof_$name.log" 2> "$out/prof_$name.err"
    echo "=== $name rc=$?"
  done
done
find "$out" -path "*prof_tcc_*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.err" -delete
