#!/bin/bash
# Round-end evidence in one go (every step under its own timeout; databases stay in /tmp, only summaries are copied):
#   gpurun_out/<tag>_bench_n1.json             python bench.py
#   gpurun_out/<tag>_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_dopri5_kernel_stats.csv   default dopri5 + adjoint call on the config-4 shard
#   gpurun_out/<tag>_dopri5_pmc_summary.csv    its MFMA / wave counters
#   gpurun_out/<tag>_wide_pmc_summary.csv      wide tile kernels (H = 64, C = 8 and H = 32, C = 16)
# Usage on the GPU box:  bash scripts/collect_profiles.sh r02
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  local db; db=$(find /tmp/prof_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $ROOT/profiles/extract_stats.py "$db" $OUT/${TAG}_${name}_kernel_stats.csv; else echo "$name: no database"; fi
}
timeout 280 python $ROOT/bench.py > $OUT/${TAG}_bench_n1.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
stats bench python $ROOT/bench.py --cpu-sample 0
stats dopri5 python $ROOT/scripts/prof_dopri5.py 2
PMC_OUT=/tmp timeout 400 bash $ROOT/scripts/pmc_passes.sh ${TAG}dopri scripts/prof_dopri5.py "mfma waves" 2 > /dev/null 2>&1
timeout 60 python $ROOT/scripts/pmc_summary.py /tmp/pmc_${TAG}dopri $OUT/${TAG}_dopri5_pmc_summary.csv > /dev/null 2>&1
for hc in "64 8" "32 16"; do
  set -- $hc
  PMC_OUT=/tmp timeout 400 bash $ROOT/scripts/pmc_passes.sh ${TAG}w$1 scripts/bench_fields.py "mfma waves" "--field linear --hidden $1 --channels $2 --variants auto --reps 1 --adjoint" > /dev/null 2>&1
  timeout 60 python $ROOT/scripts/pmc_summary.py /tmp/pmc_${TAG}w$1 /tmp/wide_$1.csv > /dev/null 2>&1
done
{ head -1 /tmp/wide_64.csv; grep -h "wide\|grad" /tmp/wide_64.csv /tmp/wide_32.csv; } > $OUT/${TAG}_wide_pmc_summary.csv 2>/dev/null
ls -la $OUT | tail -12
