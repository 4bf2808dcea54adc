#!/bin/bash
# round 6, first GPU call: 4x4x1 MFMA micro-benchmark, fresh counters for the workgroup-per-tile kernels at 4096 series,
# the per-GPU batch sweep.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
[ -x scripts/ubench/mfma_4x4 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma_4x4 scripts/ubench/mfma_4x4.hip 2>/dev/null   # (built binaries are not tracked)
./scripts/ubench/mfma_4x4 > gpurun_out/r06_mfma_4x4_ubench.txt 2>&1
python scripts/bench_strong.py --steps 20 > gpurun_out/r06_strong_before.log 2>&1
PMC_OUT=$PWD/gpurun_out bash scripts/pmc_passes.sh r06_split scripts/prof_workload.py "mfma waves fetch write" "20 4096" > gpurun_out/r06_split_pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_r06_split gpurun_out/r06_split_pmc_summary.csv
rm -rf gpurun_out/pmc_r06_split
cat gpurun_out/r06_mfma_4x4_ubench.txt gpurun_out/r06_strong_before.log gpurun_out/r06_split_pmc_summary.csv
