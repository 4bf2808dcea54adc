# Counter passes for the adaptive kernels: K4 + K4a on a config-4 shard, K4 + K4am on the
# examples' model, the bf16x3 pair on the headline.  Summaries -> gpurun_out/r03_*_pmc_summary.csv
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
bash $ROOT/scripts/pmc_passes.sh k4a scripts/prof_k4a.py "mfma waves fetch write" "seminorm 32768" > $OUT/pmc_k4a.log 2>&1
python $ROOT/scripts/pmc_summary.py /tmp/pmc_k4a $OUT/r03_dopri5_pmc_summary.csv
bash $ROOT/scripts/pmc_passes.sh k4am scripts/prof_default_mlp.py "mfma waves fetch write" "4096 seminorm" > $OUT/pmc_k4am.log 2>&1
python $ROOT/scripts/pmc_summary.py /tmp/pmc_k4am $OUT/r03_k4am_pmc_summary.csv
bash $ROOT/scripts/pmc_passes.sh bx scripts/prof_workload.py "mfma waves fetch write" "3 32768 bf16x3" > $OUT/pmc_bx.log 2>&1
python $ROOT/scripts/pmc_summary.py /tmp/pmc_bx $OUT/r03_bf16x3_pmc_summary.csv
tail -3 $OUT/pmc_k4a.log $OUT/pmc_k4am.log $OUT/pmc_bx.log
cat $OUT/r03_dopri5_pmc_summary.csv $OUT/r03_k4am_pmc_summary.csv $OUT/r03_bf16x3_pmc_summary.csv | cut -c1-400
