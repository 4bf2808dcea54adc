#!/usr/bin/env python
"""Container-side CPU baseline with the REFERENCE's own classes: torchcde.CubicSpline and
torchcde.solver._VectorField imported unmodified from /root/reference, the reference's own
hermite_cubic_coefficients_with_backward_differences, driven by the oracle's restatement of torchdiffeq (registered as
the `torchdiffeq` module: the real package is not installable here).  Runs only where /root/reference exists (the
build container, not the GPU box); writes profiles/r05_cpu_reference_container.json.

    python tests/tools/cpu_reference_baseline.py [series=4096] [threads=all]
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_golden import import_reference  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
ref = import_reference()
L, C, H = 128, 8, 32
x = make_series(n, L, C, seed=0)
z0 = torch.randn(n, H, generator=torch.Generator().manual_seed(0))
func = LinearField(H, C, scale=0.25, seed=0)


def fit():
    t0 = time.perf_counter()
    coeffs = ref.hermite_cubic_coefficients_with_backward_differences(x)
    return time.perf_counter() - t0, coeffs


def solve(coeffs, adjoint):
    X = ref.CubicSpline(coeffs)
    z = z0.clone().requires_grad_(adjoint)
    func.zero_grad()
    t0 = time.perf_counter()
    if adjoint:
        out = ref.cdeint(X, func, z, X.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
        out[:, -1].sum().backward()
    else:
        with torch.no_grad():
            ref.cdeint(X, func, z, X.interval, adjoint=False, method="rk4", options=dict(step_size=1.0))
    return time.perf_counter() - t0


_, coeffs = fit()
solve(coeffs, True)                                   # warm-up
res = {"series": n, "threads": threads, "cores_available": os.cpu_count(), "length": L, "channels": C, "hidden": H,
       "classes": "reference torchcde.CubicSpline / _VectorField / cdeint (imported from /root/reference) over the "
                  "oracle's torchdiffeq restatement (oracle/odeint.py)"}
for name, fn in (("fit", lambda: fit()[0]), ("forward", lambda: solve(coeffs, False)), ("forward_adjoint", lambda: solve(coeffs, True))):
    times = sorted(fn() for _ in range(3))
    res[name + "_s_min"] = times[0]
    res[name + "_s_median"] = statistics.median(times)
    res[name + "_series_per_s_median"] = n / statistics.median(times)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r05_cpu_reference_container.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res))
