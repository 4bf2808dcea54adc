# headline evidence after K3j: bench line, rocprof kernel stats of the same command, counter passes of the headline workload
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/tests/../bench.py > $OUT/r03_bench_n1.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
rm -rf /tmp/prof_bench
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $ROOT/bench.py --cpu-sample 0 > /tmp/prof_bench.log 2>&1
db=$(find /tmp/prof_bench -name "*.db" | head -1)
[ -n "$db" ] && python $ROOT/profiles/extract_stats.py "$db" $OUT/r03_bench_kernel_stats.csv > /dev/null
head -8 $OUT/r03_bench_kernel_stats.csv | cut -c1-160
bash $ROOT/scripts/pmc_passes.sh k3j scripts/prof_workload.py "mfma waves fetch write" "3" > $OUT/pmc_k3j.log 2>&1
python $ROOT/scripts/pmc_summary.py /tmp/pmc_k3j $OUT/r03_pmc_summary.csv
cat $OUT/r03_pmc_summary.csv | cut -c1-300
cut -c1-1500 $OUT/r03_bench_n1.json
