#!/bin/bash
# Collect hardware counters for the per-kernel microbenchmark, one rocprofv3 pass per counter group
# (a group that exceeds a block's slots hangs the profiler, so every pass is its own bounded run).
# Usage: tools/pmc.sh <outdir> <groups: comma list of fetch,write,tcc,sq,sq2,sq3> [kbench args...]
set -u
out=$1; groups=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KB_ARGS=("$@")
run() { # name counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/kbench.py --reps 2 "${KB_ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
for g in ${groups//,/ }; do
  case $g in
    fetch) run fetch FETCH_SIZE || exit 1;;
    write) run write WRITE_SIZE || exit 1;;
    tcc)   run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1;;
    sq)    run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1;;
    sq2)   run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE || exit 1;;
    sq3)   run sq3 SQ_INSTS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES || exit 1;;
  esac
done
