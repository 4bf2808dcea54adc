"""The one-layer default training call (dopri5 + adjoint: K4 forward, K4a backward) at SMALL batches -- the tanh field of
example/irregular_data.py on a Hermite-cubic control, and the affine field on a linear control -- microseconds per
attempted step.    python scripts/bench_k4a_small.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

front = sys.modules["torchcde_amd.cdeint"]
dev = torch.device("cuda", 0)
L, C, H = 128, 8, 32
for B in [int(a) for a in sys.argv[1:]] or [32, 4096]:
    x = make_series(B, L, C, seed=0).to(dev)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
    for name, X, func, extra in (
            ("tanh field, cubic control, seminorm", cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x)),
             LinearField(H, C, scale=0.25, tanh=True, seed=0).to(dev), dict(adjoint_options=dict(norm="seminorm"))),
            ("affine field, linear control, jumps, seminorm", cde.LinearInterpolation(cde.linear_interpolation_coeffs(x)),
             LinearField(H, C, scale=0.25, seed=0).to(dev), None)):
        if extra is None:
            extra = dict(options=dict(jump_t=X.grid_points), adjoint_options=dict(norm="seminorm", jump_t=X.grid_points))
        for rep in range(2):
            z = z0.detach().requires_grad_(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = cde.cdeint(X, func, z, X.interval, **extra)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            out[:, -1].sum().backward()
            torch.cuda.synchronize(); t2 = time.perf_counter()
        f, b = dict(front.last_dopri5_stats), dict(front.last_dopri5_adjoint_stats)
        print("B=%d %s: forward %.2f ms (%d attempts, %.1f us each), backward %.2f ms (%d attempts, %.1f us each) [%s]" % (
            B, name, (t1 - t0) * 1e3, f["n_accept"] + f["n_reject"], (t1 - t0) * 1e6 / max(f["launches"], 1), (t2 - t1) * 1e3,
            b["n_accept"] + b["n_reject"], (t2 - t1) * 1e6 / max(b["launches"], 1), type(out.grad_fn).__name__), flush=True)
