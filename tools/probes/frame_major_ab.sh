#!/bin/bash
# Same-box A/B of the frame-major rule of csrc/api.hip in the DIAGNOSTIC library (ADAIN_BIG_ROUNDS_X10: 60 = big layers frame by frame,
# 10000000 = every layer once over the whole batch; ADAIN_BIG_FRAME_WIDTH=0 lifts the width condition):
#   tools/probes/frame_major_ab.sh "4:0:2" "4:0:4" "5:0:2" "4:1408:2" ...      items = config:size(0 = the config's own):batch
cd "${GRAFT_REPO_ROOT:-$PWD}"
for item in "$@"; do
  IFS=: read cfg size batch <<< "$item"
  sz=""; [ "$size" != 0 ] && sz="--size $size"
  for rounds in 60 10000000; do
    out=$(ADAIN_BIG_FRAME_WIDTH=0 ADAIN_BIG_ROUNDS_X10=$rounds python bench.py --diag-lib --config $cfg $sz --batch $batch --no-cpu --no-secondary --sustain 0 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])")
    echo "config $cfg size $size batch $batch rounds_x10 $rounds: $out"
  done
done
