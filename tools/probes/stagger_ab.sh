#!/bin/bash
# A/B of ADAIN_W4_STAGGER / ADAIN_W4_PRIO on the config-2 bench (diagnostic library)
DIAG="--diag-lib"   # bench.py loads the diagnostic library itself
run() {
  echo "== $*"
  env "$@" python bench.py $DIAG --steps 20 --no-cpu --no-secondary --layers 2> /tmp/layers.txt | python -c "
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
