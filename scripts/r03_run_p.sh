# the adaptive harness after the cached-Jacobian change: config 4 forward, forward + adjoint (seminorm / default mixed norm)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { echo "== $*"; timeout 600 "$@" 2>/tmp/err.log | grep "^{"; }
{
run python bench.py --config 4 --controller local --steps 5 --warmup 1
run python bench.py --config 4 --controller local --adjoint --norm seminorm --steps 3 --warmup 1
run python bench.py --config 4 --controller local --adjoint --norm mixed --steps 2 --warmup 1
} > $OUT/r03_adaptive_bench_c.log 2>&1
cut -c1-700 $OUT/r03_adaptive_bench_c.log
