"""rk4 with the two-layer field at small batches (the split forms of K2m / K3m): forward and forward + adjoint, ms.
    python scripts/bench_k2m_small.py        (CDE_K2M_NO_SPLIT=1 / CDE_K3M_NO_SPLIT=1: one wave per tile)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import TwoLayerField, make_series  # noqa: E402

dev = "cuda"
kw = dict(method="rk4", options=dict(step_size=1.0))
for B in (64, 4096, 8192, 12288):
    x = make_series(B, 128, 8, seed=0).to(dev)
    X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
    f = TwoLayerField(32, 8, 128, seed=0).to(dev)
    z0 = torch.randn(B, 32, device=dev)

    def fwd():
        with torch.no_grad():
            cde.cdeint(X, f, z0, X.interval, **kw)

    def both():
        z = z0.clone().requires_grad_(True)
        f.zero_grad()
        cde.cdeint(X, f, z, X.interval, **kw)[:, -1].sum().backward()

    out = {}
    for name, fn in (("forward", fwd), ("forward_adjoint", both)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) * 100, 3)
    print("B", B, out)
