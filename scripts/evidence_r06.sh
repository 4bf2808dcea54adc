#!/bin/bash
# Round-6 evidence on the current build: full GPU suite (-x, as the driver runs it), smoke, achieved parity, bench line,
# rocprofv3 kernel stats of the bench command, PMC passes of the headline workload.  Summaries land in gpurun_out/; the ones to
# be judged are copied to profiles/.   usage: evidence_r06.sh [tag]   (tag: suffix of the test log, default "final")
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-final}
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/r06_gpu_tests_$TAG.log; tail -2 $OUT/r06_gpu_tests_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/r06_gpu_tests_$TAG.log
timeout 300 python tests/tools/achieved_parity.py 2>/dev/null > $OUT/r06_achieved_parity.json; head -c 400 $OUT/r06_achieved_parity.json; echo
timeout 600 python bench.py > $OUT/r06_bench_n1.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
python - <<PY
import json
d = json.load(open("$OUT/r06_bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "mfma_frac", "kernel", "kernel_ms", "traffic")})
print(d["cpu_baseline"]["value"], d["extra"]["strong_scaling_proxy_1gpu"])
print(d["extra"]["backprop_mode_adjoint_false"])
PY
bash scripts/collect_profiles.sh r06 "stats" 2>&1 | tail -2
PMC_OUT=/tmp timeout 700 bash scripts/pmc_passes.sh r06 scripts/prof_workload.py "mfma waves fetch write" 3 > /tmp/pmc.log 2>&1; tail -2 /tmp/pmc.log
timeout 60 python scripts/pmc_summary.py /tmp/pmc_r06 $OUT/r06_pmc_summary.csv; cut -c1-230 $OUT/r06_pmc_summary.csv
