#!/usr/bin/env python
"""Round 6: what control gradients cost inside the adaptive backward.  The reference's default call (dopri5 + adjoint, mixed
norm) with and without the control's coefficient tensor (and knot times) in adjoint_params:
  * config-4 shard: 32768 x 128 x 8, LinearInterpolation, linear field (K4 + K4a)
  * the examples' model on a Hermite-cubic control, 4096 x 64 x 8 (K4 + K4am)
Per case: forward ms, backward ms, attempted steps, microseconds per attempted backward step.
    python scripts/bench_adaptive_control.py [--small]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, TwoLayerField, make_series  # noqa: E402

front = sys.modules["torchcde_amd.cdeint"]
dev = torch.device("cuda", 0)
small = "--small" in sys.argv


def run(name, make_control, func, H, B, extra):
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
    res = {}
    for mode in ("parameters", "parameters+coeffs", "parameters+coeffs+knots"):
        X, coeffs, knots = make_control(mode != "parameters", mode.endswith("knots"))
        params = tuple(func.parameters()) + ((coeffs,) if mode != "parameters" else ()) + ((knots,) if mode.endswith("knots") else ())

        def once():
            z = z0.detach().requires_grad_(True)
            for p in params:
                p.grad = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = cde.cdeint(X, func, z, X.interval, adjoint_params=params, **extra)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out[:, -1].sum().backward()
            torch.cuda.synchronize()
            return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
        once()
        f, b = once()
        st = dict(front.last_dopri5_adjoint_stats)
        n = st["n_accept"] + st["n_reject"]
        res[mode] = {"dispatch": front.last_dispatch()[0].path, "forward_ms": round(f, 2), "backward_ms": round(b, 2),
                     "backward_attempts": n, "us_per_attempt": round(b * 1e3 / max(n, 1), 1)}
    print(json.dumps({name: res}))


# config-4 shard
B, L, C, H = (4096 if small else 32768), 128, 8, 32
x = make_series(B, L, C, seed=0).to(dev)


def linear_control(grad, knots):
    c = cde.linear_interpolation_coeffs(x).detach().requires_grad_(grad)
    t = torch.linspace(0, L - 1, L, device=dev).requires_grad_(True) if knots else None
    return cde.LinearInterpolation(c, t), c, t


f1 = LinearField(H, C, scale=0.25, seed=0).to(dev)
run("config4_shard_linear_field_%d" % B, linear_control, f1, H, B, {})

# the examples' model
B2, L2 = 4096, 64
x2 = make_series(B2, L2, 8, seed=1).to(dev)


def cubic_control(grad, knots):
    t = torch.linspace(0, L2 - 1, L2, device=dev).requires_grad_(True) if knots else None
    c = cde.hermite_cubic_coefficients_with_backward_differences(x2, t.detach() if knots else None).detach().requires_grad_(grad)
    return cde.CubicSpline(c, t), c, t


f2 = TwoLayerField(32, 8, 128, seed=0).to(dev)
run("example_model_cubic_%d" % B2, cubic_control, f2, 32, B2, {})
