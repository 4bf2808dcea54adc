#!/bin/bash
# Round-5 counter evidence for the adjoint=False kernels (K2 storing its stage states + K3d): the same passes as the headline's.
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
PMC_OUT=/tmp timeout 700 bash scripts/pmc_passes.sh r05bp scripts/prof_workload.py "mfma waves fetch write" "3 32768 auto backprop" > /tmp/pmc.log 2>&1; tail -3 /tmp/pmc.log
timeout 60 python scripts/pmc_summary.py /tmp/pmc_r05bp $OUT/r05_backprop_pmc_summary.csv; cut -c1-230 $OUT/r05_backprop_pmc_summary.csv
