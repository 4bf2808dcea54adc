#!/bin/bash
# round 4, call A: full GPU suite + where a small-batch K4am attempt's time goes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/r04_gpu_tests_a.log
cat $OUT/r04_gpu_tests_a.log
python scripts/bench_default_call.py 32 seminorm 2>&1 | tail -2
python scripts/bench_default_call.py 64 mixed 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
for b in 32 128; do
  rm -rf /tmp/prof_s$b
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s$b -o s$b -- python $ROOT/scripts/prof_default_mlp.py $b seminorm > /tmp/prof_s$b.log 2>&1
  db=$(find /tmp/prof_s$b -name "*.db" | head -1)
  python $ROOT/profiles/extract_stats.py "$db" $OUT/r04_k4am_${b}_seminorm_kernel_stats_before.csv
  head -8 $OUT/r04_k4am_${b}_seminorm_kernel_stats_before.csv | cut -c1-200
done
