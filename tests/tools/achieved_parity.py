"""The parity figures ACHIEVED at the benchmark size (SURVEY 8(d) asks for allclose(rtol 1e-4, atol 1e-6) on trajectories and
rtol 1e-3 on gradients; VERDICT round 4: "state the achieved figure"): BASELINE configs[2] -- 32768 series, L = 128, C = 8,
H = 32, rk4 step 1 -- K2 forward + K3p adjoint (and adjoint=False: K2 + K3d) against the float64 oracle, every element.
    python tests/tools/achieved_parity.py > profiles/r05_achieved_parity.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from oracle import cde as oracle_cde, interp as oracle_interp  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

B, L, C, H = 32768, 128, 8, 32
torch.set_num_threads(min(16, os.cpu_count() or 1))
x = make_series(B, L, C, seed=0)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0))
kw = dict(method="rk4", options=dict(step_size=1.0))
result = {"workload": "BASELINE configs[2]: 32768 x 128 x 8, H = 32, rk4 step 1.0, loss = z_T.sum()"}
for adjoint in (True, False):
    f64 = LinearField(H, C, torch.float64, scale=0.25, seed=0)
    f32 = LinearField(H, C, torch.float32, scale=0.25, seed=0)
    outs, gzs, outs32 = [], [], []
    for lo in range(0, B, 2048):
        Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x[lo:lo + 2048].double()))
        zo = z0[lo:lo + 2048].double().requires_grad_(True)
        o = oracle_cde.cdeint(Xo, f64, zo, Xo.interval, adjoint=adjoint, **kw)
        o[:, -1].sum().backward()
        outs.append(o.detach()); gzs.append(zo.grad)
        if adjoint:                                       # the float32 CPU oracle: what float32 arithmetic itself costs
            with torch.no_grad():
                X32 = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x[lo:lo + 2048]))
                outs32.append(oracle_cde.cdeint(X32, f32, z0[lo:lo + 2048], X32.interval, adjoint=False, **kw))
    ref_out, ref_gz = torch.cat(outs), torch.cat(gzs)
    X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x.cuda()))
    func = LinearField(H, C, scale=0.25, seed=0).cuda()
    z = z0.cuda().requires_grad_(True)
    out = cde.cdeint(X, func, z, X.interval, adjoint=adjoint, **kw)
    out[:, -1].sum().backward()

    def figures(got, want):
        got, want = got.detach().double().cpu(), want.double()
        err = (got - want).abs()
        return {"max_abs_err": err.max().item(), "max_abs_value": want.abs().max().item(),
                "max_err_over_(1e-6 + 1e-4 |ref|)": (err / (1e-6 + 1e-4 * want.abs())).max().item(),
                "max_err_over_largest_entry": (err.max() / want.abs().max()).item()}
    key = "adjoint_true_K2_K3p" if adjoint else "adjoint_false_K2_K3d"
    result[key] = {"trajectory_z_T": figures(out[:, -1], ref_out[:, -1]), "dL_dz0": figures(z.grad, ref_gz),
                   "dL_dW": figures(func.linear.weight.grad, f64.linear.weight.grad),
                   "dL_db": figures(func.linear.bias.grad, f64.linear.bias.grad)}
    if adjoint:
        result["float32_cpu_oracle_trajectory_z_T"] = figures(torch.cat(outs32)[:, -1], ref_out[:, -1])
print(json.dumps(result, indent=1))
