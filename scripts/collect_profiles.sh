#!/bin/bash
# Round-end evidence in one go (every step under its own timeout; databases stay in /tmp, only summaries are copied to
# gpurun_out/, from where the ones to be judged are committed under profiles/):
#   <tag>_gpu_tests.log                         python -m pytest tests -m gpu -q
#   <tag>_bench_n1.json                         python bench.py
#   <tag>_bench_kernel_stats.csv                rocprofv3 --kernel-trace --stats of `python bench.py --cpu-sample 0`
#   <tag>_k4am_{32,4096}_seminorm_kernel_stats.csv   the example model's default call (K4 forward, K4am backward)
#   <tag>_phase_k4am_{32,4096}.log              where an attempt's time goes (instrumented library)
#   <tag>_k4am_pmc_summary.csv                  MFMA-busy / wait / instruction counters of the K4am kernels at 4096 series
#   <tag>_fuzz_seed61.log                       AUTO kernels vs generic / step-wise on drawn configurations
#   <tag>_multigpu_harness_rccl_1rank.log       bench.py --config 5 / --config 4 --adjoint, local vs shared controller, one RCCL rank
# Usage on the GPU box:  bash scripts/collect_profiles.sh r04 [steps...]     (default: all steps)
set -u
TAG=${1:-r04}
STEPS=${2:-"tests bench stats k4am phase pmc fuzz dist"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  local db; db=$(find /tmp/prof_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $ROOT/profiles/extract_stats.py "$db" $OUT/${TAG}_${name}_kernel_stats.csv; else echo "$name: no database"; tail -3 /tmp/prof_$name.log; fi
}
for step in $STEPS; do
  case $step in
    tests) (cd $ROOT && timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/${TAG}_gpu_tests.log; cat $OUT/${TAG}_gpu_tests.log | tail -2) ;;
    bench) (cd $ROOT && timeout 600 python bench.py > $OUT/${TAG}_bench_n1.json 2> /tmp/bench.err || tail -5 /tmp/bench.err; python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "mfma_frac", "kernel_ms", "traffic")})
e = d["extra"]["other_configs"]
print({k: round(v, 1) if isinstance(v, float) else v for k, v in e.items() if "example_model" in k})
print(d["extra"]["strong_scaling_proxy_1gpu"])
PY
) ;;
    stats) stats bench python $ROOT/bench.py --cpu-sample 0 ;;
    k4am) stats k4am_32_seminorm python $ROOT/scripts/prof_default_mlp.py 32 seminorm
          stats k4am_4096_seminorm python $ROOT/scripts/prof_default_mlp.py 4096 seminorm ;;
    phase) (cd $ROOT && CDE_PHASE_TRACE=1 timeout 300 python scripts/phase_trace.py k4am 32 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_phase_k4am_32.log
            CDE_PHASE_TRACE=1 timeout 300 python scripts/phase_trace.py k4am 4096 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_phase_k4am_4096.log; head -6 $OUT/${TAG}_phase_k4am_4096.log) ;;
    pmc) PMC_OUT=/tmp timeout 500 bash $ROOT/scripts/pmc_passes.sh ${TAG}k4am scripts/prof_default_mlp.py "mfma waves" "4096 seminorm" > /dev/null 2>&1
         timeout 60 python $ROOT/scripts/pmc_summary.py /tmp/pmc_${TAG}k4am $OUT/${TAG}_k4am_pmc_summary.csv; cat $OUT/${TAG}_k4am_pmc_summary.csv | cut -c1-220 ;;
    dist) (cd $ROOT && { for mode in local shared; do CDE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config 5 --controller $mode --steps 3 --warmup 1 2>/dev/null; done
                        for mode in local shared; do CDE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config 5 --norm seminorm --controller $mode --steps 3 --warmup 1 2>/dev/null; done
                        for mode in local shared; do CDE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config 4 --adjoint --norm seminorm --controller $mode --steps 5 --warmup 2 2>/dev/null; done; } > $OUT/${TAG}_multigpu_harness_rccl_1rank.log
           python - <<PY
import json
for line in open("$OUT/${TAG}_multigpu_harness_rccl_1rank.log"):
    d = json.loads(line)
    print(d["config"].get("controller"), d["config"].get("adjoint_norm"), d["metric"][:60], round(d["ms_per_step"], 1), "ms/step")
PY
) ;;
    fuzz) (cd $ROOT && timeout 600 python tests/tools/fuzz_variants.py --cases 100 --seed 61 2>&1 | tail -6 > $OUT/${TAG}_fuzz_seed61.log; cat $OUT/${TAG}_fuzz_seed61.log) ;;
  esac
done
ls -la $OUT | tail -14
