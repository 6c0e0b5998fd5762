#!/bin/bash
# A/B of ADAIN_W4_STAGGER / ADAIN_W4_PRIO on the config-2 bench (diagnostic library)
export ADAIN_HIP_LIB=$PWD/applied-image-processing_amd/libadain_hip_diag.so
run() {
  echo "== $*"
  env "$@" python bench.py --steps 20 --no-cpu --no-secondary --layers 2> /tmp/layers.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
  grep layer /tmp/layers.txt | awk '{printf "%s ", $6}'; echo
}
run ADAIN_W4_STAGGER=0 ADAIN_W4_PRIO=1
run ADAIN_W4_STAGGER=6 ADAIN_W4_PRIO=1
run ADAIN_W4_STAGGER=12 ADAIN_W4_PRIO=1
run ADAIN_W4_STAGGER=20 ADAIN_W4_PRIO=1
run ADAIN_W4_STAGGER=0 ADAIN_W4_PRIO=0
run ADAIN_W4_STAGGER=12 ADAIN_W4_PRIO=0
run ADAIN_W4_STAGGER=0 ADAIN_W4_PRIO=2
run ADAIN_W4_STAGGER=12 ADAIN_W4_PRIO=2
run ADAIN_W4_STAGGER=0 ADAIN_W4_PRIO=3
run ADAIN_W4_STAGGER=12 ADAIN_W4_PRIO=3
run ADAIN_W4_STAGGER=0 ADAIN_W4_PRIO=1
