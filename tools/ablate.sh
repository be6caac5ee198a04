#!/bin/bash
for a in 0 1 2 4 8 3 6 7 15 9 10 12; do
  echo "ablate $a"; MGRT_ABLATE=$a timeout -k 10 120 python tools/kbench.py --reps 5 --only "sweep x2 " | grep sweep || exit 1
done
