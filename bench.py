#!/usr/bin/env python
"""bench.py -- series/sec (fwd+adjoint) for cdeint RK4, batch=32768, L=128, C=8, H=32 (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --config 4 [--controller local|shared] [--adjoint] [--gpus N]     BASELINE configs[3] (see run_adaptive_config)
    python bench.py --config 5 [--method dopri5|rk4] [--controller local|shared] [--gpus N]   BASELINE configs[4] (see run_logode_config)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic series already resident in HBM:
``cdeint(X, func, z0, X.interval, method='rk4', options={'step_size': 1.0})`` (K2: 127 RK4 steps) followed
by ``z_T.sum().backward()`` (K3: continuous-adjoint reverse sweep, gradients for z0, W, b).  This is
BASELINE.json configs[2] ("same as [1] with adjoint=True backprop through solver, 1 MI355X"), the
configuration the metric is quoted on.  Coefficients are fitted once outside the timed region (K1; the
reference treats it as dataset pre-processing) and its rate is reported separately under "extra".

Multi-GPU: series are independent, so the batch shards across ranks with NO data-path collective; the only
communication is the all-reduce of the 8,448 parameter gradients, included in the timed step when N > 1.
``--scaling strong`` (default; what BASELINE.json asks for: ONE 32768-series job on N GPUs) gives every rank
32768 / N series -- below 16384 series per GPU cdeint switches to the workgroup-per-tile kernels (rk4_split.hip),
which keep every SIMD busy down to 16 series per CU.  ``--scaling weak`` keeps 32768 series per rank.
``python bench.py --gpus N`` without a torchrun environment re-executes itself under torch.distributed.run with N
ranks on 127.0.0.1.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, L, C, H = 32768, 128, 8, 32
N_EVAL = 4 * (L - 1)
# algorithmic work per series (restated in DESIGN.md)
FLOP_FWD = N_EVAL * (2 * H * (H * C) + 2 * H * C)                                     # 8.58 MFLOP
FLOP_ADJ = N_EVAL * ((2 * H * H * C + 2 * H * C) + 2 * H * C + 2 * H * C * H + 2 * H * C * H + H * C)   # 25.6 MFLOP
PEAK_F32_MFMA_TFLOPS = 157.3
# What K3j (csrc/rk4_mfma.hip, the default adjoint kernel of the affine field) EXECUTES per evaluation: the shared Jacobian
# J = sum_c dX_c W_c replaces the two GEMMs f = W (z (x) dX), a^T df/dz = W^T (a (x) dX) of the count above by one GEMM of
# the same size plus two H x H matrix-vector products on the vector pipe.  Matrix pipe: J (+ bias rows) and dL/dW.
FLOP_ADJ_K3J_MFMA = N_EVAL * ((2 * H * H * C + 2 * H * C) + 2 * H * C * H)            # 16.9 MFLOP
FLOP_ADJ_K3J_VALU = N_EVAL * (2 * 2 * H * H + H * C)                                   # 2.2 MFLOP
PEAK_HBM_GBS = 8000.0


def adjoint_hbm_traffic(kernel_substring):
    """HBM bytes per launch of the dominant kernel, read from the newest profiles/r*_pmc_summary.csv that holds a row for it
    (separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, scripts/pmc_passes.sh + scripts/pmc_summary.py;
    FETCH_SIZE doubled: gfx950 tallies the 128-byte requests of 16 B/lane reads at 64 bytes, MI355X_MICROARCH.md "HBM").
    Returns (bytes or None, source string or None)."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")), reverse=True):
        try:
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel_substring in row.get("kernel", "") and row.get("fetch_KB_raw") and row.get("write_KB_raw"):
                        fetch, write = float(row["fetch_KB_raw"]) * 1e3, float(row["write_KB_raw"]) * 1e3
                        return int(2 * fetch + write), (
                            "profiles/%s, row %s: 2 x %.1f MB FETCH_SIZE (gfx950 half-count correction for 16 B/lane reads) + "
                            "%.1f MB WRITE_SIZE, separate --pmc passes" % (os.path.basename(path), row["kernel"], fetch / 1e6,
                                                                          write / 1e6))
        except (OSError, ValueError):
            continue
    return None, None


def make_workload(device, seed, n=None, first=0, count=None):
    """The SURVEY 8(d) workload: n series generated on the CPU from `seed` (identical bits on every rank and on the
    CPU baseline), rows [first, first + count) moved to `device`."""
    from helpers import LinearField, make_series
    n = B if n is None else n
    count = n if count is None else count
    x = make_series(n, L, C, seed=seed)[first:first + count].contiguous().to(device)
    func = LinearField(H, C, scale=0.25, seed=0).to(device)
    z0 = torch.randn(n, H, generator=torch.Generator().manual_seed(seed))[first:first + count].contiguous().to(device)
    return x, func, z0


def _log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C stdio when its
# first communicator comes up), so file descriptor 1 points at stderr for the whole run and the result line is written to
# the saved descriptor at the very end.
_REAL_STDOUT = None


def capture_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)                    # whatever C code buffered for "stdout" goes to stderr now
    except OSError:
        pass
    fd = 1 if _REAL_STDOUT is None else _REAL_STDOUT
    os.write(fd, (line + "\n").encode())


def cpu_baseline(max_sample, budget_s=20.0):
    """The oracle (torch-CPU restatement of the reference path) on a bounded sample of the same workload, timed per
    BASELINE.md section 3: 1 warm-up + 3 timed runs of the same sample, min and median reported, `value` = median.

    The CPU gets its best shot: the thread count is chosen among {8, 16, 32, 64} (capped by the cores this process
    may run on -- cgroup/affinity aware) by a 1024-series probe, and the sample is sized from that probe so that the
    four runs together take about ``budget_s`` seconds (large batches amortise eager-op overheads: bigger is
    fairer to the CPU; the per-series rate at 8192 series is within a few per cent of the full-batch rate)."""
    from oracle import cde as oracle_cde, interp as oracle_interp
    from helpers import LinearField, make_series
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    func = LinearField(H, C, scale=0.25, seed=0)

    def run(n):
        x = make_series(n, L, C, seed=0)
        z0 = torch.randn(n, H, generator=torch.Generator().manual_seed(0))
        X = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x))
        z = z0.clone().requires_grad_(True)
        func.zero_grad()
        t0 = time.perf_counter()
        out = oracle_cde.cdeint(X, func, z, X.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
        out[:, -1].sum().backward()
        return time.perf_counter() - t0

    best_threads, best_time = 1, float("inf")
    for threads in (8, 16, 32, 64):
        if threads > avail:
            break
        torch.set_num_threads(threads)
        run(128)                              # warm-up for this pool size
        dt = run(1024)
        _log("cpu baseline probe: 1024 series, %d threads: %.2f s" % (threads, dt))
        if dt < best_time:
            best_threads, best_time = threads, dt
    if best_time == float("inf"):             # fewer than 8 cores available
        best_threads = max(1, avail)
        torch.set_num_threads(best_threads)
        run(128)
        best_time = run(1024)
    torch.set_num_threads(best_threads)
    sample = int(min(max_sample, max(1024, 1024 * budget_s / 4 / best_time))) // 1024 * 1024
    run(sample)                               # warm-up at the timed size
    times = sorted(run(sample) for _ in range(3))
    return {"value": sample / times[1], "unit": "series/s", "cores": best_threads, "kind": "port",
            "best_value": sample / times[0], "runs_s": times,
            "sample": "oracle (torch-CPU restatement of reference CubicSpline + _VectorField + torchdiffeq rk4/"
                      "adjoint) on %d of the %d series, L=%d, fwd+adjoint: 1 warm-up + 3 timed runs, median %.2f s "
                      "(min %.2f s), the fastest of {8,16,32,64} threads (%d cores available); container-side figure "
                      "with the reference's own CubicSpline/_VectorField classes: profiles/r05_cpu_reference_container"
                      ".json" % (sample, B, L, times[1], times[0], avail)}


def strong_scaling_proxy(cde, x, func, z0, full_ms):
    """What ONE GPU of an N-GPU strong-scaling run does, measured on this GPU: forward + adjoint on 32768/N series
    (N = 2, 4, 8) against the full batch.  Wall clock over 20 steps after 3 warm-ups.  The real run adds one 33 KB
    gradient all-reduce per step: its cost is measured here on a ONE-RANK RCCL group (the call path and launch of
    `allreduce_gradients`, not the 8-rank ring latency -- no multi-GPU node is available to this run) and added to the
    shard time for `speedup_at_N_gpus_incl_allreduce`."""
    out = {"batch_32768_ms": full_ms}
    params = list(func.parameters())
    allreduce_ms = measured_allreduce_ms(params)
    out["allreduce_33KB_1rank_rccl_ms"] = allreduce_ms
    for n_gpus in (2, 4, 8):
        b = 32768 // n_gpus
        if b > x.size(0):
            continue
        X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x[:b].contiguous()))
        zz = z0[:b].contiguous()

        def once():
            z = zz.detach().requires_grad_(True)
            for p in params:
                p.grad = None
            cde.cdeint(X, func, z, X.interval, method="rk4", options={"step_size": 1.0})[:, -1].sum().backward()
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            once()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        out["batch_%d_ms" % b] = ms
        out["speedup_at_%d_gpus_compute_only" % n_gpus] = full_ms / ms
        if allreduce_ms is not None:
            out["speedup_at_%d_gpus_incl_allreduce" % n_gpus] = full_ms / (ms + allreduce_ms)
    return out


def measured_allreduce_ms(params):
    """`distributed.allreduce_gradients` of the 8,448 parameter gradients on a one-rank RCCL process group, ms per call
    (stream time over 50 calls after 5 warm-ups); None when no group can be formed here."""
    import socket
    import torch.distributed as dist
    from torchcde_amd.distributed import allreduce_gradients
    created = False
    try:
        if not dist.is_initialized():
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            import datetime
            dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                    timeout=datetime.timedelta(seconds=60))
            created = True
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        for _ in range(5):
            allreduce_gradients(params)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(50):
            allreduce_gradients(params)
        ev[1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e3
        return max(ev[0].elapsed_time(ev[1]) / 50, wall)          # the larger of stream time and host time per call
    except Exception as err:                                      # no RCCL here: the proxy stays compute-only
        _log("all-reduce timing skipped: %r" % (err,))
        return None
    finally:
        if created:
            dist.destroy_process_group()


def bf16x3_variant(cde, X, func, z0, steps=10):
    """The same headline step through variant="bf16x3" (csrc/rk4_bf16x3.hip: the weight GEMMs on the bf16 matrix pipe, every
    float32 operand split into three bf16 pieces -- float32 accuracy, same parity bars; NOT the reported `value`, which
    stays on the exact-f32 kernels): ms per forward solve and per forward + adjoint step, wall clock over `steps`."""
    kw = dict(method="rk4", options={"step_size": 1.0}, variant="bf16x3")
    params = list(func.parameters())

    def fwd():
        with torch.no_grad():
            cde.cdeint(X, func, z0, X.interval, **kw)

    def both():
        z = z0.detach().requires_grad_(True)
        for p in params:
            p.grad = None
        cde.cdeint(X, func, z, X.interval, **kw)[:, -1].sum().backward()
    out = {}
    for name, fn in (("forward_ms", fwd), ("forward_adjoint_ms", both)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
    out["series_per_s"] = z0.size(0) / (out["forward_adjoint_ms"] * 1e-3)
    return out


def backprop_mode(cde, X, func, z0, steps=10):
    """The same workload through cdeint(..., adjoint=False) (reference solver.py:144; README.md:103 "the faster mode"):
    K2 storing its stage states + K3d, the reverse-mode sweep of csrc/rk4_backprop.hip -- the gradient of the DISCRETE
    solve.  NOT the reported `value` (BASELINE's metric is forward + adjoint).  ms per forward + backward step, wall clock
    over `steps`, and the two kernels' own durations from HIP events on the launching stream."""
    front = sys.modules["torchcde_amd.cdeint"]            # its `event_log` attribute is this thread's
    kw = dict(method="rk4", options={"step_size": 1.0}, adjoint=False)
    params = list(func.parameters())

    def both():
        z = z0.detach().requires_grad_(True)
        for p in params:
            p.grad = None
        cde.cdeint(X, func, z, X.interval, **kw)[:, -1].sum().backward()
    gc.collect()                       # a full collection of a torch process takes ~50 ms: not inside a 66 ms timed region,
    for _ in range(5):                 # and before the warm-up (the GPU idles meanwhile and drops its clocks)
        both()
    torch.cuda.synchronize()
    front.event_log = []
    t0 = time.perf_counter()
    for _ in range(steps):
        both()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    log, front.event_log = front.event_log, None
    fwd = [a.elapsed_time(b) for kind, a, b in log if kind == "forward"]
    bwd = [a.elapsed_time(b) for kind, a, b in log if kind == "backprop"]
    B = z0.size(0)
    bwd_ms = sum(bwd) / max(len(bwd), 1)
    if os.environ.get("CDE_BENCH_DEBUG"):
        _log("backprop_mode: wall %.3f fwd %s bwd %s dispatch %s" % (wall, [round(v, 2) for v in fwd], [round(v, 2) for v in bwd],
                                                                   front.last_dispatch()))
    flop = B * N_EVAL * (2 * 2 * H * H * C + 2 * H * H + 2 * H * C)       # J's GEMM + dL/dW on the matrix pipe, J^T kb + dL/db
    return {"forward_backward_ms": wall, "series_per_s": B / (wall * 1e-3),
            "forward_with_stage_stores_kernel_ms": sum(fwd) / max(len(fwd), 1), "backward_kernel_ms": bwd_ms,
            "backward_executed_tflops": flop / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else None,
            "stage_state_bytes": B * (N_EVAL) * 32 * 4}


def other_fields(cde, X, z0, device, reps=3):
    """Same workload with the non-linear vector fields of the reference's examples (outside the timed region, not part
    of `value`): Linear -> tanh (example/irregular_data.py) and Linear -> relu -> Linear -> tanh, width 128
    (example/time_series_classification.py); and the linear field with 64 hidden units (wide tile kernels).  ms per solve,
    best of `reps` single solves after one warm-up."""
    class TwoLayer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear1, self.linear2 = torch.nn.Linear(H, 128), torch.nn.Linear(128, H * C)

        def forward(self, t, z):
            return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], H, C)

    from helpers import LinearField
    torch.manual_seed(0)
    fields = {"tanh": LinearField(H, C, scale=1.0, tanh=True, seed=0).to(device), "two_layer": TwoLayer().to(device),
              # the same control with 64 hidden units: the wide tile kernels (csrc/rk4_wide.hip)
              "linear_hidden64": LinearField(64, C, scale=0.5, seed=0).to(device)}
    starts = {"linear_hidden64": torch.randn(z0.size(0), 64, generator=torch.Generator().manual_seed(1)).to(device)}
    out = {}
    for name, func in fields.items():
        start = starts.get(name, z0)
        # (adjoint=False -- reverse mode through the solver's steps, README.md:103 -- is fused for the two-layer field too)
        for mode in ("forward", "forward_adjoint") + (("forward_backprop_adjoint_false",) if name == "two_layer" else ()):
            def once():
                if mode == "forward":
                    with torch.no_grad():
                        cde.cdeint(X, func, start, X.interval, method="rk4", options={"step_size": 1.0})
                else:
                    z = start.detach().requires_grad_(True)
                    cde.cdeint(X, func, z, X.interval, method="rk4", options={"step_size": 1.0},
                               adjoint=mode == "forward_adjoint")[:, -1].sum().backward()
            once()
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(reps):                  # best of `reps` single solves (each one synchronised)
                t0 = time.perf_counter()
                once()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out["%s_%s_ms" % (name, mode)] = best * 1e3
    return out


def other_configs(cde, device, reps=3):
    """BASELINE configs[3] (one GPU's shard) and configs[4] on this GPU, outside the timed region (ms, best of `reps` single
    runs after one warm-up):
      configs[3]: 32768 x 128 x 8, linear_interpolation_coeffs + LinearInterpolation, dopri5 (rtol 1e-4, atol 1e-6,
                  jump_t = the knots), linear func; forward, and forward + adaptive adjoint backward (K4 / K4a)
      configs[4]: 32768 x 512 x 3 -> depth-3 logsignatures over windows of 8 (65 x 14) -> LinearInterpolation ->
                  two-layer field (hidden size 8 as example/logsignature_example.py:22, width 128), rk4, adjoint."""
    from helpers import LinearField, make_series

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(reps):                      # best of `reps` single runs (each one synchronised)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    out = {}
    x = make_series(B, L, C, seed=0).to(device)
    X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
    func = LinearField(H, C, scale=0.25, seed=0).to(device)          # as scripts/bench_dopri5.py: 129 accepted steps
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(device)
    kw = dict(method="dopri5", rtol=1e-4, atol=1e-6, options=dict(jump_t=X.grid_points))

    def dopri_forward():
        with torch.no_grad():
            cde.cdeint(X, func, z0, X.interval, **kw)

    def dopri_adjoint():
        z = z0.detach().requires_grad_(True)
        cde.cdeint(X, func, z, X.interval, **kw)[:, -1].sum().backward()

    front = sys.modules["torchcde_amd.cdeint"]

    def dopri_adjoint_seminorm():
        z = z0.detach().requires_grad_(True)
        cde.cdeint(X, func, z, X.interval, adjoint_options=dict(norm="seminorm", jump_t=X.grid_points), **kw)[:, -1].sum().backward()

    out["config4_shard_dopri5_forward_ms"] = timed(dopri_forward)
    # the reference's default adjoint (torchdiffeq's MIXED norm: the parameter-gradient blocks drive the step size -- in
    # float32 their error estimate sits at the noise floor of the batch sums, hence the rejected attempts) and the same call
    # with adjoint_options=dict(norm="seminorm")
    out["config4_shard_dopri5_forward_adjoint_ms"] = timed(dopri_adjoint)
    out["config4_shard_dopri5_adjoint_attempts"] = {k: v for k, v in front.last_dopri5_adjoint_stats.items()
                                                    if k in ("n_accept", "n_reject")}
    out["config4_shard_dopri5_forward_adjoint_seminorm_ms"] = timed(dopri_adjoint_seminorm)

    class TwoLayer(torch.nn.Module):
        def __init__(self, hidden, channels):
            super().__init__()
            self.hidden, self.channels = hidden, channels
            self.linear1, self.linear2 = torch.nn.Linear(hidden, 128), torch.nn.Linear(128, hidden * channels)

        def forward(self, t, z):
            return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], self.hidden, self.channels)

    gen = torch.Generator().manual_seed(1)
    raw = (torch.randn(B, 512, 3, generator=gen) * 0.1).cumsum(1)
    raw[..., 0] = torch.linspace(0, 1, 512)
    raw = raw.to(device)
    torch.manual_seed(0)
    field = TwoLayer(8, 14).to(device)
    z8 = torch.randn(B, 8, generator=gen).to(device)
    state = {}

    def transform():
        state["X"] = cde.LinearInterpolation(cde.linear_interpolation_coeffs(cde.logsig_windows(raw, 3, 8.0)))

    def solve():
        z = z8.detach().requires_grad_(True)
        Xl = state["X"]
        cde.cdeint(Xl, field, z, Xl.interval, method="rk4", options={"step_size": 1.0})[:, -1].sum().backward()

    out["config5_logsig_transform_ms"] = timed(transform)
    out["config5_two_layer_forward_adjoint_ms"] = timed(solve)
    out["config5_series_per_s"] = B / ((out["config5_logsig_transform_ms"] + out["config5_two_layer_forward_adjoint_ms"]) * 1e-3)

    # config 5 as the reference's example RUNS it (example/logsignature_example.py builds the same CDEFunc and calls cdeint
    # without a method: dopri5 + adjoint): K4 forward, K4am backward -- one run each after a warm-up
    def solve_default(extra):
        z = z8.detach().requires_grad_(True)
        Xl = state["X"]
        cde.cdeint(Xl, field, z, Xl.interval, **extra)[:, -1].sum().backward()

    def once(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    solve_default({})
    out["config5_default_method_forward_adjoint_ms"] = once(lambda: solve_default({}))
    out["config5_default_method_attempts"] = {"forward": dict(front.last_dopri5_stats),
                                              "backward": {k: v for k, v in front.last_dopri5_adjoint_stats.items()
                                                           if k in ("n_accept", "n_reject")}}
    out["config5_default_method_seminorm_forward_adjoint_ms"] = once(lambda: solve_default(dict(adjoint_options=dict(norm="seminorm"))))

    # the same pipeline at hidden size 32 (round 6: 32 hidden units x 14 channels, the upper unit groups of the output layer read
    # from L2): rk4 forward + adjoint, and the reference's default call
    torch.manual_seed(0)
    field32 = TwoLayer(32, 14).to(device)
    z32 = torch.randn(B, 32, generator=gen).to(device)

    def solve32(extra):
        z = z32.detach().requires_grad_(True)
        Xl = state["X"]
        cde.cdeint(Xl, field32, z, Xl.interval, **extra)[:, -1].sum().backward()

    out["config5_hidden32_two_layer_forward_adjoint_ms"] = timed(lambda: solve32(dict(method="rk4", options={"step_size": 1.0})))
    out["config5_hidden32_dispatch"] = front.last_dispatch()[0].path
    solve32({})
    out["config5_hidden32_default_method_forward_adjoint_ms"] = once(lambda: solve32({}))

    # the example model's own training call (example/time_series_classification.py:83-86: cdeint(X, CDEFunc, z0, X.interval),
    # 4096 series of the headline workload; round 2: 1.0 s forward + 34 s backward step-wise)
    torch.manual_seed(0)
    model = TwoLayer(H, C).to(device)
    xs = make_series(4096, L, C, seed=0).to(device)
    Xs = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(xs))
    zs0 = torch.randn(4096, H, generator=torch.Generator().manual_seed(0)).to(device)
    # ... and at the examples' own batch size (batch_size=32, example/time_series_classification.py:149): the eight-waves-
    # per-tile attempt kernel with the factor reduction + R step in one launch
    zb0 = zs0[:32].contiguous()
    Xb = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(xs[:32].contiguous()))
    for tag, extra in (("", {}), ("_seminorm", dict(adjoint_options=dict(norm="seminorm")))):
        zb = zb0.detach().requires_grad_(True)
        cde.cdeint(Xb, model, zb, Xb.interval, **extra)
        fwd_ms = once(lambda: cde.cdeint(Xb, model, zb, Xb.interval, **extra))
        res = cde.cdeint(Xb, model, zb, Xb.interval, **extra)
        out["example_model_batch32_default_call%s_forward_ms" % tag] = fwd_ms
        out["example_model_batch32_default_call%s_backward_ms" % tag] = once(lambda: res[:, -1].sum().backward())
        st = front.last_dopri5_adjoint_stats
        out["example_model_batch32_default_call%s_us_per_attempt" % tag] = (
            out["example_model_batch32_default_call%s_backward_ms" % tag] * 1e3 / max(st.get("n_accept", 0) + st.get("n_reject", 0), 1))
    for tag, extra in (("", {}), ("_seminorm", dict(adjoint_options=dict(norm="seminorm")))):
        zs = zs0.detach().requires_grad_(True)
        cde.cdeint(Xs, model, zs, Xs.interval, **extra)        # warm-up of the forward kernels
        fwd_ms = once(lambda: cde.cdeint(Xs, model, zs, Xs.interval, **extra))
        res = cde.cdeint(Xs, model, zs, Xs.interval, **extra)
        assert type(res.grad_fn).__name__ == "_FusedMlpDopri5Backward"
        out["example_model_default_call%s_forward_ms" % tag] = fwd_ms
        out["example_model_default_call%s_backward_ms" % tag] = once(lambda: res[:, -1].sum().backward())
        out["example_model_default_call%s_backward_attempts" % tag] = {
            k: v for k, v in front.last_dopri5_adjoint_stats.items() if k in ("n_accept", "n_reject")}
    # ... at 8192 series: the shared-tile forms of K4 / K4am over two rounds of workgroups (up to 12288 series)
    x8 = make_series(8192, L, C, seed=0).to(device)
    X8 = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x8))
    z8 = torch.randn(8192, H, generator=torch.Generator().manual_seed(0)).to(device).requires_grad_(True)
    semi = dict(adjoint_options=dict(norm="seminorm"))
    cde.cdeint(X8, model, z8, X8.interval, **semi)
    out["example_model_8192_default_call_seminorm_forward_ms"] = once(lambda: cde.cdeint(X8, model, z8, X8.interval, **semi))
    res = cde.cdeint(X8, model, z8, X8.interval, **semi)
    out["example_model_8192_default_call_seminorm_backward_ms"] = once(lambda: res[:, -1].sum().backward())
    st = front.last_dopri5_adjoint_stats
    out["example_model_8192_default_call_seminorm_us_per_attempt"] = (
        out["example_model_8192_default_call_seminorm_backward_ms"] * 1e3 / max(st.get("n_accept", 0) + st.get("n_reject", 0), 1))
    # ... and the same model under rk4 (K2m + K3m, the eight-wave sweep below 24576 series): forward + adjoint
    for nb, Xr, zr in ((64, None, None), (4096, Xs, zs0)):
        if Xr is None:
            Xr = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(xs[:nb].contiguous()))
            zr = zs0[:nb].contiguous()

        def rk4_step():
            z = zr.detach().requires_grad_(True)
            cde.cdeint(Xr, model, z, Xr.interval, method="rk4", options=dict(step_size=1.0))[:, -1].sum().backward()
        rk4_step()
        out["example_model_rk4_forward_adjoint_%d_series_ms" % nb] = min(once(rk4_step) for _ in range(3))
    return out


def run_adaptive_config(args, cde, device, rank, world, distributed, share_gpu):
    """--config 4: BASELINE configs[3] -- 32768 series PER GPU (262,144 over 8), linear_interpolation_coeffs +
    LinearInterpolation, dopri5 (rtol 1e-4, atol 1e-6, jump_t = the knots), linear func; one step = one adaptive forward
    solve of this rank's shard (`--adjoint`: + the default adjoint backward and the gradient all-reduce).
    `--controller local`: every rank runs its own step controller (no collective at all in the forward solve);
    `--controller shared`: torchcde_amd.distributed.shared_step_control -- the pending error sums of EVERY attempted step
    are all-reduced over RCCL, so all shards take the step sequence of the unsharded 32768 x N batch (torchdiffeq's
    controller is batch-global).  With CDE_BENCH_FORCE_DIST=1 on a 1-GPU box the shared mode measures the cost of that
    per-attempt all-reduce on one rank."""
    import contextlib
    import torch.distributed as dist
    from helpers import LinearField, make_series
    from torchcde_amd.distributed import allreduce_gradients, shared_step_control
    front = sys.modules["torchcde_amd.cdeint"]
    n = 32768
    x = make_series(n, L, C, seed=rank).to(device)
    X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
    func = LinearField(H, C, scale=0.25, seed=0).to(device)
    z0 = torch.randn(n, H, generator=torch.Generator().manual_seed(rank)).to(device)
    kw = dict(method="dopri5", rtol=1e-4, atol=1e-6, options=dict(jump_t=X.grid_points))
    if args.norm == "seminorm":
        kw["adjoint_options"] = dict(norm="seminorm", jump_t=X.grid_points)
    params = list(func.parameters())
    shared = args.controller == "shared"
    if shared and not distributed:
        raise SystemExit("--controller shared needs a process group (N > 1, or CDE_BENCH_FORCE_DIST=1 on one GPU)")

    def step():
        scope = shared_step_control(n * world) if shared else contextlib.nullcontext()
        with scope:
            if args.adjoint:
                z = z0.detach().requires_grad_(True)
                for p in params:
                    p.grad = None
                cde.cdeint(X, func, z, X.interval, **kw)[:, -1].sum().backward()
                if distributed:
                    allreduce_gradients(params)
            else:
                with torch.no_grad():
                    cde.cdeint(X, func, z0, X.interval, **kw)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    if rank == 0:
        fwd, bwd = dict(front.last_dopri5_stats), dict(front.last_dopri5_adjoint_stats) if args.adjoint else {}
        attempts = fwd.get("launches", 0) + bwd.get("launches", 0)
        emit(json.dumps({
            "metric": "series/sec, dopri5 adaptive cdeint %s, 32768 series per GPU, L=128 C=8 H=32 (BASELINE configs[3])"
                      % ("forward + adjoint" if args.adjoint else "forward"),
            "value": n * world * args.steps / elapsed, "unit": "series/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (LAUNCHER CHECK: ranks share cuda:0 over gloo)" if share_gpu else ""),
            "config": {"workload": "BASELINE configs[3]: linear_interpolation_coeffs + LinearInterpolation, dopri5 rtol 1e-4 "
                                   "atol 1e-6 jump_t = knots, linear func; %d series in total" % (n * world),
                       "controller": args.controller, "adjoint": bool(args.adjoint), "adjoint_norm": args.norm,
                       "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": "batch-sharded x%d; %s" % (world, "one 16-byte all-reduce per attempted step (forward)"
                                                                 if shared else "no collective in the solve")},
            "extra": {"forward_steps": fwd, "backward_steps": {k: v for k, v in bwd.items() if k in ("n_accept", "n_reject", "launches")},
                      "kernel_launches_per_step": attempts,
                      "us_per_attempted_step": elapsed / args.steps * 1e6 / max(attempts, 1)}}))


def run_logode_config(args, cde, device, rank, world, distributed, share_gpu):
    """--config 5: BASELINE configs[4] -- the log-ODE pipeline on 32768 series per GPU: raw (32768, 512, 3) -> depth-3
    logsignatures over windows of 8 (K5, 65 x 14) -> LinearInterpolation -> the two-layer field of
    example/logsignature_example.py:21-23 (hidden size 8, width 128), forward + adjoint, gradient all-reduce.
    `--method dopri5` is the call as the reference's example makes it (no method: K4 + K4am); `--method rk4` the fixed-step
    solve (K2m + K3m).  One step = transform + solve + backward of this rank's shard."""
    import torch.distributed as dist
    from helpers import TwoLayerField
    from torchcde_amd.distributed import allreduce_gradients
    front = sys.modules["torchcde_amd.cdeint"]
    n = 32768
    gen = torch.Generator().manual_seed(1 + rank)
    raw = (torch.randn(n, 512, 3, generator=gen) * 0.1).cumsum(1)
    raw[..., 0] = torch.linspace(0, 1, 512)
    raw = raw.to(device)
    hidden = args.hidden
    field = TwoLayerField(hidden, 14, 128, seed=0).to(device)
    z8 = torch.randn(n, hidden, generator=gen).to(device)
    params = list(field.parameters())
    solver = dict(method="rk4", options={"step_size": 1.0}) if args.method == "rk4" else {}
    if args.method != "rk4" and args.norm == "seminorm":
        solver["adjoint_options"] = dict(norm="seminorm")
    times = {"transform": 0.0}

    import contextlib
    from torchcde_amd.distributed import shared_step_control
    shared = args.controller == "shared" and args.method != "rk4"
    if shared and not distributed:
        raise SystemExit("--controller shared needs a process group (N > 1, or CDE_BENCH_FORCE_DIST=1 on one GPU)")

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        Xl = cde.LinearInterpolation(cde.linear_interpolation_coeffs(cde.logsig_windows(raw, 3, 8.0)))
        ev[1].record()
        z = z8.detach().requires_grad_(True)
        for p in params:
            p.grad = None
        # --controller shared: ONE step controller for the whole sharded batch (K4 / K4am with the two-layer field, round 4)
        with (shared_step_control(n * world) if shared else contextlib.nullcontext()):
            cde.cdeint(Xl, field, z, Xl.interval, **solver)[:, -1].sum().backward()
        if distributed:
            allreduce_gradients(params)
        return ev

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    events = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    times["transform"] = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    if rank == 0:
        extra = {"logsig_transform_ms": times["transform"]}
        if args.method != "rk4":
            extra["forward_steps"] = dict(front.last_dopri5_stats)
            extra["backward_steps"] = {k: v for k, v in front.last_dopri5_adjoint_stats.items()
                                       if k in ("n_accept", "n_reject", "launches")}
        emit(json.dumps({
            "metric": "series/sec, log-ODE pipeline (depth-3 logsignature windows + two-layer field, %s, fwd+adjoint), "
                      "32768 x 512 x 3 per GPU (BASELINE configs[4])" % args.method,
            "value": n * world * args.steps / elapsed, "unit": "series/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (LAUNCHER CHECK: ranks share cuda:0 over gloo)" if share_gpu else ""),
            "config": {"workload": "BASELINE configs[4]: logsig_windows(depth 3, window 8) -> LinearInterpolation -> "
                                   "Linear(%d,128)-relu-Linear(128,%d)-tanh field, method %s, adjoint=True; %d series in total"
                                   % (hidden, hidden * 14,
                                      "dopri5 (the reference example's default call)" if args.method != "rk4" else "rk4 step 1",
                                      n * world),
                       "hidden_channels": hidden, "dispatch": front.last_dispatch()[0].path,
                       "method": args.method, "adjoint_norm": args.norm, "controller": "shared" if shared else "local",
                       "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": "batch-sharded x%d, one gradient all-reduce per step" % world},
            "extra": extra}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, choices=(3, 4, 5), default=3,
                    help="3 (default): the headline metric, BASELINE configs[2]; 4: configs[3] (sharded adaptive dopri5 solve); "
                         "5: configs[4] (log-ODE pipeline)")
    ap.add_argument("--controller", choices=("local", "shared"), default="local",
                    help="--config 4 / 5: one step controller per rank, or ONE for the whole sharded batch (all-reduce per attempt)")
    ap.add_argument("--adjoint", action="store_true", help="--config 4: also time the default adjoint backward")
    ap.add_argument("--method", choices=("dopri5", "rk4"), default="dopri5", help="--config 5: the solver")
    ap.add_argument("--hidden", type=int, choices=(8, 32), default=8,
                    help="--config 5: hidden size of the two-layer field (8: example/logsignature_example.py:22; 32: the 32 x 16 tiles)")
    ap.add_argument("--norm", choices=("mixed", "seminorm"), default="mixed",
                    help="adaptive adjoint: torchdiffeq's default mixed norm or adjoint_options=dict(norm='seminorm')")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: 32768 series in total (32768/N per GPU); weak: 32768 series per GPU")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="series in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # not under torchrun: launch the N ranks ourselves (one process per GPU, RCCL over xGMI, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    capture_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        _log("note: --gpus %d but WORLD_SIZE=%d; the launcher's world size is what runs and what is reported"
             % (args.gpus, world))
    global B
    B_TOTAL = B
    if args.scaling == "strong" and args.config == 3:
        if B_TOTAL % world:
            raise SystemExit("strong scaling needs 32768 %% n_gpus == 0, got n_gpus=%d" % world)
        B = B_TOTAL // world                      # per-rank batch; FLOP/byte bookkeeping below is per rank
    # CDE_BENCH_FORCE_DIST=1 exercises the RCCL code path with a single rank (used to validate it on a 1-GPU box).
    # CDE_BENCH_SHARE_GPU=1 is a LAUNCHER check for 1-GPU boxes only: every rank uses cuda:0 and the process group is
    # gloo (RCCL refuses two ranks on one device) -- it validates self-launch / sharding / reporting, not a number.
    distributed = world > 1 or os.environ.get("CDE_BENCH_FORCE_DIST") == "1"
    share_gpu = os.environ.get("CDE_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:              # CDE_BENCH_FORCE_DIST=1 without a launcher: a one-rank process group
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            try:                                          # "nccl" is RCCL on ROCm; bind the communicator to this GPU
                dist.init_process_group(backend="nccl", device_id=device)
            except TypeError:
                dist.init_process_group(backend="nccl")

    import torchcde_amd as cde
    front_events = sys.modules["torchcde_amd.cdeint"]      # its `event_log` attribute is this thread's
    from torchcde_amd.distributed import allreduce_gradients
    cde.load()
    if args.config != 3:
        runner = run_adaptive_config if args.config == 4 else run_logode_config
        runner(args, cde, device, rank, world, distributed, share_gpu)
        if distributed:
            dist.destroy_process_group()
        return

    _log("rank %d/%d building workload (%d series on this rank, %s scaling)" % (rank, world, B, args.scaling))
    if args.scaling == "strong":                  # this rank's shard of THE 32768-series job
        x, func, z0 = make_workload(device, seed=0, n=B_TOTAL, first=rank * B, count=B)
    else:
        x, func, z0 = make_workload(device, seed=rank)

    # K1 outside the timed region, timed on its own
    torch.cuda.synchronize()
    for _ in range(2):
        coeffs = cde.hermite_cubic_coefficients_with_backward_differences(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        coeffs = cde.hermite_cubic_coefficients_with_backward_differences(x)
    ev[1].record()
    torch.cuda.synchronize()
    fit_ms = ev[0].elapsed_time(ev[1]) / 10
    # K0 (missing-value fill) on the survey's irregular workload: 10 % of the entries missing
    x_nan = x.masked_fill(torch.rand(x.shape, device=device) < 0.1, float("nan"))
    cde.linear_interpolation_coeffs(x_nan)
    ev[0].record()
    for _ in range(5):
        cde.linear_interpolation_coeffs(x_nan)
    ev[1].record()
    torch.cuda.synchronize()
    fill_ms = ev[0].elapsed_time(ev[1]) / 5
    del x_nan
    X = cde.CubicSpline(coeffs)
    t = X.interval
    params = list(func.parameters())

    def step():
        z = z0.detach().requires_grad_(True)
        for p in params:
            p.grad = None
        out = cde.cdeint(X, func, z, t, method="rk4", options={"step_size": 1.0})
        out[:, -1].sum().backward()
        if distributed:
            allreduce_gradients(params)
        return out

    _log("hermite fit %.3f ms; warm-up" % fit_ms)
    # the interpreter's own full collection (~50 ms in a torch process) is not part of the step being timed: run it now, BEFORE
    # the warm-up (the GPU idles meanwhile and needs the warm-up steps to come back to its clocks)
    gc.collect()
    gc.freeze()                                  # (what survives is permanent: later collections only look at new objects)
    prewarm = int(os.environ.get("CDE_BENCH_PREWARM", "5"))
    for _ in range(prewarm):   # untimed, before the contract's W warm-up steps: back to full clocks
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _log("timing %d steps" % args.steps)

    front_events.event_log = []                                   # HIP events around the K2 / K3 C-ABI calls
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log, front_events.event_log = front_events.event_log, None

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    fwd_ms = [a.elapsed_time(b) for kind, a, b in log if kind == "forward"]
    adj_ms = [a.elapsed_time(b) for kind, a, b in log if kind == "adjoint"]
    fwd_avg = sum(fwd_ms) / max(len(fwd_ms), 1)
    adj_avg = sum(adj_ms) / max(len(adj_ms), 1)

    _log("timed region done: %.3f s" % elapsed)
    if rank == 0:
        total_series = B * world * args.steps
        value = total_series / elapsed
        split = B <= 16384                      # CDE_SPLIT_MAX_BATCH: workgroup-per-tile kernels below, K2/K3 above
        jacobian = cde.get_option("k3_form") != 1
        pair = jacobian and cde.get_option("k3_waves") != 1                    # K3p: chain + helper wave per tile (the default)
        kernel = ("rk4_adjoint_split8" if split else "rk4_adjoint_jacobian_pair" if pair else
                  "rk4_adjoint_jacobian" if jacobian else "rk4_adjoint_mfma")
        # `achieved` = the flop the kernel's formulation EXECUTES per launch / its average duration.  The default kernels of
        # the affine field take f AND a^T df/dz from the shared Jacobian J = sum_c dX_c W_c: per series and evaluation
        # 33,280 flop on the matrix pipe (J with its bias rows, dL/dW) + 4,352 on the vector pipe (two H x H matrix-vector
        # products, dL/db) = 37,632 -- not the 50,432 of the reference's three-GEMM formulation, which is kept
        # beside it as `reference_formulation_*` (the rate a kernel of that formulation would need for the same duration).
        flop_exec = B * (FLOP_ADJ_K3J_MFMA + FLOP_ADJ_K3J_VALU) if jacobian else B * FLOP_ADJ
        flop_mfma = B * FLOP_ADJ_K3J_MFMA if jacobian else B * (FLOP_ADJ - N_EVAL * (H * C + 4 * H * C))
        sec = adj_avg * 1e-3
        achieved = flop_exec / sec / 1e12 if adj_avg > 0 else 0.0
        mfma_tf = flop_mfma / sec / 1e12 if adj_avg > 0 else 0.0
        ref_tf = B * FLOP_ADJ / sec / 1e12 if adj_avg > 0 else 0.0
        traffic, traffic_source = adjoint_hbm_traffic(kernel)
        if B != (4096 if split else 32768):
            traffic, traffic_source = None, None          # the counter passes were run at 32768 series (K3j / K3) and 4096 (K3s)
        result = {
            "metric": "series/sec (fwd+adjoint) for cdeint RK4, batch=32k L=128 C=8 H=32",
            "value": value,
            "unit": "series/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (LAUNCHER CHECK: all ranks share cuda:0 over gloo -- not a measurement)"
                                   if share_gpu else ""),
            "config": {"workload": "BASELINE configs[2]: Hermite-cubic control, linear func Linear(32,256), RK4 "
                                   "step 1.0 (127 steps), cdeint forward + adjoint=True backward; %s"
                                   % ("ONE 32768-series batch sharded over the GPUs" if args.scaling == "strong"
                                      else "32768 series per GPU"),
                       "batch_per_gpu": B, "length": L, "input_channels": C, "hidden_channels": H,
                       "global_batch": B * world, "parallelism": "batch-sharded x%d, no data-path collective, one "
                                                                  "33 KB gradient all-reduce per step" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "mfma_frac": mfma_tf / PEAK_F32_MFMA_TFLOPS, "mfma_tflops": mfma_tf,
                         "kernel": kernel + {"rk4_adjoint_split8": " (K3s)", "rk4_adjoint_jacobian": " (K3j)",
                                             "rk4_adjoint_jacobian_pair": " (K3p)", "rk4_adjoint_mfma": " (K3)"}[kernel],
                         "kernel_ms": adj_avg, "executed_flop_per_launch": flop_exec,
                         "formulation": ("shared Jacobian: 33,280 MFMA + 4,352 VALU flop per series and evaluation" if jacobian
                                         else "three GEMMs: 50,432 flop per series and evaluation (SURVEY 8(d))"),
                         "reference_formulation_flop": B * FLOP_ADJ, "reference_formulation_tflops": ref_tf,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": B * 12704},
            "extra": {
                # methodology since round 5 (disclosed here so that lines of different rounds can be compared): `prewarm`
                # untimed steps BEFORE the contract's --warmup steps, and one gc.collect() + gc.freeze() before them
                "untimed_prewarm_steps_before_warmup": prewarm, "gc_frozen_before_timing": True,
                "forward_kernel_ms": fwd_avg,
                "forward_tflops": B * FLOP_FWD / (fwd_avg * 1e-3) / 1e12 if fwd_avg > 0 else None,
                "forward_only_series_per_s": B / (fwd_avg * 1e-3) if fwd_avg > 0 else None,
                "end_to_end_fit_fwd_adjoint_series_per_s": B * world / ((fit_ms + elapsed / args.steps * 1e3) * 1e-3),
                "hermite_fit_ms": fit_ms,
                "hermite_fit_series_per_s": B / (fit_ms * 1e-3),
                "hermite_fit_hbm_gbs": B * 20352 / (fit_ms * 1e-3) / 1e9,
                "hermite_fit_hbm_frac_of_8TBs": B * 20352 / (fit_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "missing_value_fill_ms_10pct_nan": fill_ms,
                "missing_value_fill_series_per_s": B / (fill_ms * 1e-3),
            },
        }
        if world == 1:
            result["extra"]["strong_scaling_proxy_1gpu"] = strong_scaling_proxy(cde, x, func, z0, elapsed / args.steps * 1e3)
        if world == 1:
            result["extra"]["bf16x3_variant"] = bf16x3_variant(cde, X, func, z0)
            result["extra"]["backprop_mode_adjoint_false"] = backprop_mode(cde, X, func, z0)
            result["extra"]["other_fields"] = other_fields(cde, X, z0, device)
            result["extra"]["other_configs"] = other_configs(cde, device)
        if world == 1 and args.cpu_sample > 0:
            result["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, B))
        emit(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
