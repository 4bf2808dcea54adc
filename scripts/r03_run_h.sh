# round-3 GPU batch: multi-GPU harness on one rank (RCCL all-reduce cost of the shared controller), config 5 default method
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { echo "== $*"; timeout 600 "$@" 2>/tmp/err.log | tail -1; tail -2 /tmp/err.log | grep -v "^\[bench\]" ; }
{
run python bench.py --config 4 --steps 5 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller local --steps 5 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller shared --steps 5 --warmup 1
run python bench.py --config 4 --adjoint --norm seminorm --steps 3 --warmup 1
CDE_BENCH_FORCE_DIST=1 run python bench.py --config 4 --controller shared --adjoint --norm seminorm --steps 3 --warmup 1
run python bench.py --config 4 --adjoint --steps 2 --warmup 1
run python bench.py --config 5 --method rk4 --steps 5 --warmup 1
run python bench.py --config 5 --method dopri5 --norm seminorm --steps 2 --warmup 1
run python bench.py --config 5 --method dopri5 --steps 1 --warmup 1
} > $OUT/r03_multigpu_harness_1rank.log 2>&1
cat $OUT/r03_multigpu_harness_1rank.log | cut -c1-1500
