#!/bin/bash
# A/B of two builds of the library in ONE GPU call (boxes differ by several per cent, so numbers from
# different calls do not compare): tools/ab.sh <libA.so> <libB.so> [kbench sizes...]
# Prints the cycle time at 513^3 and the sweep timings at the given sizes for both, alternating.
set -u
a=$1; b=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for l in "$a" "$b"; do
    echo "== $l (round $rep)"
    MGRT_LIB=$PWD/$l timeout -k 10 100 python3 bench.py --no-cpu --steps 30 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cycle %.4f ms  roofline kernel %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_s']*1e6))" || exit 1
    for n in "$@"; do
      MGRT_LIB=$PWD/$l timeout -k 10 60 python3 tools/kbench.py --n $n --reps 30 --only "sweep" 2>/dev/null | grep -v "correct+" | sed "s/^/  $n: /" || exit 1
    done
  done
done
