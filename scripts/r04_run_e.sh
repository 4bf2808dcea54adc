#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_layer_default_call or edge_cases_two_layer or fused_equals_stepwise" 2>&1 | tail -4
python scripts/bench_default_call.py 32 seminorm 2>&1 | tail -1
python scripts/bench_default_call.py 32 mixed 2>&1 | tail -1
python scripts/bench_default_call.py 256 seminorm 2>&1 | tail -1
CDE_K4AM_NO_SMALL_REDUCE=1 python scripts/bench_default_call.py 256 seminorm 2>&1 | tail -1
CDE_PHASE_TRACE=1 python scripts/phase_trace.py k4am 32 2>&1 | tail -25
