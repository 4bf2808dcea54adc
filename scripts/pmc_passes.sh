#!/bin/bash
# Counter passes for the roofline evidence (one --pmc set per run, kernel-trace only: see the gpurun rules).
# Usage on the GPU box:  bash scripts/pmc_passes.sh <tag> [workload.py] [passes] [workload args]   -> /tmp/pmc_<tag>/*.db
set -u
TAG=${1:-r01}
WORKLOAD=${2:-scripts/prof_workload.py}
PASSES=${3:-"mfma waves fetch write"}
WARGS=${4:-3}
OUT=${PMC_OUT:-/tmp}/pmc_$TAG     # the .db files are large: summarise on the box (scripts/pmc_summary.py), copy the CSVs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 180 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python $GRAFT_REPO_ROOT/$WORKLOAD $WARGS > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
for pass in $PASSES; do
  case $pass in
    mfma) run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES ;;
    waves) run waves SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ;;
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
  esac
done
ls -la $OUT
