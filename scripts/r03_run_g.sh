cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
stats() { local name=$1; shift; rm -rf /tmp/prof_$name; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1; local db; db=$(find /tmp/prof_$name -name "*.db" | head -1); if [ -n "$db" ]; then python $ROOT/profiles/extract_stats.py "$db" $OUT/r03_${name}_kernel_stats.csv; else echo "$name: no db"; tail -5 /tmp/prof_$name.log; fi; }
cd $ROOT
(python scripts/bench_default_call.py 4096; python scripts/bench_default_call.py 4096 seminorm; python scripts/bench_default_call.py 32768 seminorm; python scripts/bench_dopri5_adjoint.py 32768) > $OUT/r03_adaptive_bench_b.log 2>&1
cat $OUT/r03_adaptive_bench_b.log
cd /tmp
stats k4am_4096 python $ROOT/scripts/prof_default_mlp.py 4096 seminorm
stats k4a_32768 python $ROOT/scripts/prof_dopri5.py 1
head -8 $OUT/r03_k4am_4096_kernel_stats.csv | cut -c1-200
head -8 $OUT/r03_k4a_32768_kernel_stats.csv | cut -c1-200
cd $ROOT; python -m pytest tests -m gpu -q -k "dopri5 or two_layer or adaptive" 2>&1 | tail -5
