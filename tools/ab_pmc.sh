#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the sweeps for two builds of the library in one GPU call:
# tools/ab_pmc.sh <out dir> <libA.so> <libB.so> [kbench args...]
set -u
out=$1; a=$2; b=$3; shift 3
cd "$GRAFT_REPO_ROOT" || exit 1
for l in "$a" "$b"; do
  tag=$(basename "$l" .so)
  MGRT_LIB=$PWD/$l bash tools/pmc.sh "$out/$tag" fetch,write "$@" || exit 1
  echo "== $l"; python3 tools/pmc_summary.py "$out/$tag"
done
