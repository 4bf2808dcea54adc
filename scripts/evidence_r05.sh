set -u
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r05 "stats" 2>&1 | tail -5
PMC_OUT=/tmp timeout 700 bash scripts/pmc_passes.sh r05 scripts/prof_workload.py "mfma waves fetch write" 3 > /tmp/pmc.log 2>&1; tail -3 /tmp/pmc.log
timeout 60 python scripts/pmc_summary.py /tmp/pmc_r05 gpurun_out/r05_pmc_summary.csv; cat gpurun_out/r05_pmc_summary.csv | cut -c1-260
