#!/bin/bash
# timing-only ablations of the one-tile F(4,3) x F(2,3) kernel against the one-tile product kernel (diagnostic library)
for d in 13 14 10 6; do
  echo "== ADAIN_W4_DIAG=$d"
  ADAIN_W4_DIAG=$d ADAIN_W4_PERSIST=0 python tools/probes/diag_ab.py
done
