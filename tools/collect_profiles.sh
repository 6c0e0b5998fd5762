#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (one kernel trace, three separate --pmc passes; never combined with
# sys/hip/hsa tracing).  Usage: tools/collect_profiles.sh <tag>  -> gpurun_out/prof_<tag>_{trace,fetch,write,sq}
set -o pipefail
tag="$1"; root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out"
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
run() {   # name seconds rocprof-args...
  name="$1"; secs="$2"; shift 2
  echo "=== $name"
  timeout -k 10 "$secs" rocprofv3 "$@" --output-format csv -d "$out/prof_${tag}_$name" -- python3 "$root/bench.py" --no-cpu --no-secondary $BENCH_ARGS \
      > "$out/prof_${tag}_$name.log" 2> "$out/prof_${tag}_$name.err"
  rc=$?; echo "=== $name rc=$rc"; tail -n 1 "$out/prof_${tag}_$name.log" | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
}
BENCH_ARGS="--steps 10 --warmup 2" run trace 300 --kernel-trace --stats
BENCH_ARGS="--steps 5 --warmup 1" run fetch 300 --pmc FETCH_SIZE --kernel-trace
BENCH_ARGS="--steps 5 --warmup 1" run write 300 --pmc WRITE_SIZE --kernel-trace
BENCH_ARGS="--steps 5 --warmup 1" run sq 300 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace
# keep the merge-back small: only the csv files
find "$out" -path "*prof_${tag}_*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.err" -delete
du -sh "$out"/prof_${tag}_* | tail -n 8
