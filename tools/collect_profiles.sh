#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (kernel traces and SEPARATE --pmc passes; never combined with
# sys/hip/hsa tracing; the program goes straight after `--`).
# Usage: tools/collect_profiles.sh <tag> [part ...]    parts: trace pmc configs waits all_kernels  (default: all but waits)
#   -> gpurun_out/prof_<tag>_{trace,fetch,write,sq,cfg3_trace,cfg4_trace,cfg5_trace,all_kernels}
set -o pipefail
tag="$1"; shift; parts="${*:-trace pmc configs all_kernels}"
root="${GRAFT_REPO_ROOT:-$PWD}"; out="$root/gpurun_out"
mkdir -p "$out"
# the binary and sources on this box must be the ones of the manifest's commit (tools/profile_manifest.py --write, run before gpurun)
python3 "$root/tools/profile_manifest.py" --check || exit 2
cd /tmp; export TMPDIR=/tmp
run() {   # name seconds program rocprof-args...   (BENCH_ARGS = the program's arguments)
  name="$1"; secs="$2"; prog="$3"; shift 3
  echo "=== $name"
  timeout -k 10 "$secs" rocprofv3 "$@" --output-format csv -d "$out/prof_${tag}_$name" -- python3 "$root/$prog" $BENCH_ARGS \
      > "$out/prof_${tag}_$name.log" 2> "$out/prof_${tag}_$name.err"
  rc=$?; echo "=== $name rc=$rc"; tail -n 1 "$out/prof_${tag}_$name.log" | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
}
for part in $parts; do
  case $part in
    trace)   # config 2 WITH the secondary leg: the frame-sized pixel kernels of the post-pass are in the same trace
      BENCH_ARGS="--no-cpu --steps 10 --warmup 2 --sustain 2" run trace 300 bench.py --kernel-trace --stats ;;
    pmc)
      BENCH_ARGS="--no-cpu --no-secondary --steps 5 --warmup 1 --sustain 2" run fetch 300 bench.py --pmc FETCH_SIZE --kernel-trace
      BENCH_ARGS="--no-cpu --no-secondary --steps 5 --warmup 1 --sustain 2" run write 300 bench.py --pmc WRITE_SIZE --kernel-trace
      BENCH_ARGS="--no-cpu --no-secondary --steps 5 --warmup 1 --sustain 2" run sq 300 bench.py --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace ;;
    configs)
      for cfg in 3 4 5; do
        BENCH_ARGS="--config $cfg --no-cpu --no-secondary --steps 5 --warmup 1 --sustain 2" run cfg${cfg}_trace 300 bench.py --kernel-trace --stats
      done ;;
    waits)   # wave-state and L2 hit/miss counters of configs 2 and 3 (is the larger map's extra L2-miss traffic free?)
      for cfg in 2 3; do
        BENCH_ARGS="--config $cfg --no-cpu --no-secondary --steps 3 --warmup 1 --sustain 2" run cfg${cfg}_waits 300 bench.py --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace
        BENCH_ARGS="--config $cfg --no-cpu --no-secondary --steps 3 --warmup 1 --sustain 2" run cfg${cfg}_tcc 300 bench.py --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace
      done ;;
    all_kernels)
      BENCH_ARGS="" run all_kernels 300 tools/all_kernels.py --kernel-trace --stats ;;
  esac
done
# keep the merge-back small: only the csv files
find "$out" -path "*prof_${tag}_*" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.err" -delete
du -sh "$out"/prof_${tag}_* | tail -n 12
