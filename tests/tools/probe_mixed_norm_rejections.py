"""Why does torchdiffeq's DEFAULT adjoint norm reject three attempts out of four on config 4?  (CPU, oracle only.)

The oracle (oracle/odeint.py, torchdiffeq's odeint_adjoint restated) runs the reference's default training call on a
1024-series, L = 32 version of BASELINE configs[3] (LinearInterpolation, jump_t = knots, rtol 1e-4, atol 1e-6) in float32
AND float64, under the default mixed norm and under "seminorm", and prints the accepted / rejected step counts of the
backward solve.  If the rejections were a float32 noise-floor effect the float64 run would not show them; it does.

    python tests/tools/probe_mixed_norm_rejections.py > profiles/r04_mixed_norm_rejections_probe.log
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import LinearField, make_series          # noqa: E402
from oracle import cde as oracle_cde, interp as oracle_interp, odeint as oracle_ode      # noqa: E402


def run(dtype, adjoint_options, B=1024, L=32, C=8, H=32):
    log = []
    original = oracle_ode._Dopri5.integrate

    def integrate(self, t):
        out = original(self, t)
        log.append(self)
        return out

    oracle_ode._Dopri5.integrate = integrate
    try:
        x = make_series(B, L, C, seed=0).to(dtype)
        z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dtype).requires_grad_(True)
        f = LinearField(H, C, dtype, scale=0.25, seed=0)
        X = oracle_interp.LinearPath(x)
        kw = dict(options=dict(jump_t=X.grid_points))
        if adjoint_options is not None:
            kw["adjoint_options"] = dict(adjoint_options, jump_t=X.grid_points)
        out = oracle_cde.cdeint(X, f, z0, X.interval, adjoint=True, method="dopri5", **kw)
        out[:, -1].sum().backward()
    finally:
        oracle_ode._Dopri5.integrate = original
    fwd, bwd = log[0], log[1:]
    return (fwd.n_accept, fwd.n_reject), (sum(s.n_accept for s in bwd), sum(s.n_reject for s in bwd))


if __name__ == "__main__":
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    print("oracle (torchdiffeq restated), config-4 settings on 1024 series, L = 32, jump_t = knots, rtol 1e-4, atol 1e-6")
    print("%-10s %-12s %-22s %-22s" % ("dtype", "adjoint norm", "forward acc + rej", "backward acc + rej"))
    for dtype in (torch.float32, torch.float64):
        for name, opts in (("mixed", None), ("seminorm", dict(norm="seminorm"))):
            fwd, bwd = run(dtype, opts)
            print("%-10s %-12s %-22s %-22s" % (str(dtype).replace("torch.", ""), name, "%d + %d" % fwd, "%d + %d" % bwd))
