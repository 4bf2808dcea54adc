"""dopri5 forward of one-layer fields at small batches (the split forms of K4): ms per solve.
    python scripts/bench_k4_tanh_small.py          (CDE_K4_NO_SPLIT=1: one wave per tile)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

dev = "cuda"
for B in (64, 4096):
    x = make_series(B, 128, 8, seed=0).to(dev)
    Xc = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
    Xl = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
    z0 = torch.randn(B, 32, device=dev)
    cases = (("tanh field, cubic control, default tolerances", Xc, LinearField(32, 8, scale=0.5, tanh=True, seed=0).to(dev), {}),
             ("identity field, linear control, config-4 settings", Xl, LinearField(32, 8, scale=0.25, seed=0).to(dev),
              dict(rtol=1e-4, atol=1e-6, options=dict(jump_t=Xl.grid_points))))
    for name, X, f, kw in cases:
        with torch.no_grad():
            for _ in range(2):
                out = cde.cdeint(X, f, z0, X.interval, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = cde.cdeint(X, f, z0, X.interval, **kw)
            torch.cuda.synchronize()
        print("B", B, name, "ms", round((time.perf_counter() - t0) * 200, 3), "finite", bool(torch.isfinite(out).all()))
