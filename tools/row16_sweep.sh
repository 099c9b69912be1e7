#!/bin/bash
# for every width C the 16-lane tiling of H = 16 C and H = 16 C - 8 (forced) against the planner's own choice: tools/row16_probe.py
for C in 8 10 12 14 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30; do
  H1=$((16 * C)); H2=$((16 * C - 8))
  python tools/row16_probe.py $H1 $H2 && AGX_PHMM_FORCE_C=$C python tools/row16_probe.py $H1 $H2 || exit 1
done
