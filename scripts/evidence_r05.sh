#!/bin/bash
# Round-5 evidence in one go (on the GPU box): achieved parity figures, bench line, rocprofv3 kernel stats of the bench
# command, PMC passes of the headline workload.  Summaries land in gpurun_out/, the ones to be judged are copied to profiles/.
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tests/tools/achieved_parity.py 2>/dev/null > $OUT/r05_achieved_parity.json; head -c 600 $OUT/r05_achieved_parity.json; echo
timeout 600 python bench.py > $OUT/r05_bench_n1.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
python - <<PY
import json
d = json.load(open("$OUT/r05_bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "mfma_frac", "kernel", "kernel_ms", "traffic")})
print(d["cpu_baseline"])
print(d["extra"]["strong_scaling_proxy_1gpu"])
print(d["extra"]["backprop_mode_adjoint_false"])
PY
bash scripts/collect_profiles.sh r05 "stats" 2>&1 | tail -2
PMC_OUT=/tmp timeout 700 bash scripts/pmc_passes.sh r05 scripts/prof_workload.py "mfma waves fetch write" 3 > /tmp/pmc.log 2>&1; tail -2 /tmp/pmc.log
timeout 60 python scripts/pmc_summary.py /tmp/pmc_r05 $OUT/r05_pmc_summary.csv; cut -c1-230 $OUT/r05_pmc_summary.csv
