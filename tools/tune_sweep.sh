#!/bin/bash
# development: time the fused sweep for each tile shape compiled under -DMGRT_TUNE
for n in "$@"; do
for c in ${CFGS:-0 1 2 3 4 5 6 7 8 9 10 11 12}; do
  echo -n "n=$n cfg $c: "; MGRT_SWEEP_CFG=$c timeout -k 10 120 python tools/kbench.py --n $n --reps 20 --only "sweep x" | grep -v omega | awk '{printf "%s %s %s us | ", $1, $2, $3} END {print ""}' || exit 1
done
done
