cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { echo "== $*"; timeout 600 "$@" 2>/tmp/err.log | grep "^{"; }
{
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller local --steps 5 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller shared --steps 5 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller shared --adjoint --norm seminorm --steps 3 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 5 --method rk4 --steps 5 --warmup 1
} > $OUT/r03_multigpu_harness_rccl_1rank.log 2>&1
cat $OUT/r03_multigpu_harness_rccl_1rank.log | cut -c1-900
