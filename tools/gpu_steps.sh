#!/bin/bash
# Runs GPU steps one after another on the gpurun box; a step that fails with an ordinary error does not stop the sequence, a step
# that is killed by its timeout does (no further GPU step after a hang).  Usage: tools/gpu_steps.sh name:seconds:command ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; secs="${rest%%:*}"; cmd="${rest#*:}"
  echo "=== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
  rc=$?
  echo "=== $name rc=$rc"; tail -n 3 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping"; exit 1; fi
done
exit 0
