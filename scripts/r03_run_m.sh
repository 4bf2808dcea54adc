cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
stats() { local name=$1; shift; rm -rf /tmp/prof_$name; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1; local db; db=$(find /tmp/prof_$name -name "*.db" | head -1); if [ -n "$db" ]; then python $ROOT/profiles/extract_stats.py "$db" $OUT/r03_${name}_kernel_stats.csv > /dev/null; head -5 $OUT/r03_${name}_kernel_stats.csv | cut -c1-150; else echo "$name: no db"; tail -5 /tmp/prof_$name.log; fi; }
stats k4a_seminorm python $ROOT/scripts/prof_k4a.py seminorm
stats k4a_mixed python $ROOT/scripts/prof_k4a.py mixed
stats k4am_4096_seminorm python $ROOT/scripts/prof_default_mlp.py 4096 seminorm
stats k4am_32768_seminorm python $ROOT/scripts/prof_default_mlp.py 32768 seminorm
