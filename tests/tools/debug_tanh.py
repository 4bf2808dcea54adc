import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native
from oracle import cde as oracle_cde, interp as oracle_interp
from helpers import LinearField, make_series
DEV = torch.device("cuda:0")
for scale in (0.25, 1.5):
    B, L, C, H = 203, 24, 8, 32
    x = make_series(B, L, C, torch.float32, seed=31)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    func = LinearField(H, C, torch.float32, scale=scale, tanh=True, seed=9)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(10))
    t_out = torch.tensor([0., 7.5, 23.])
    f64 = LinearField(H, C, torch.float64, scale=scale, tanh=True, seed=9)
    f64.linear.weight.data.copy_(func.linear.weight.double()); f64.linear.bias.data.copy_(func.linear.bias.double())
    with torch.no_grad():
        ref = oracle_cde.cdeint(oracle_interp.CubicPath(coeffs.double()), f64, z0.double(), t_out.double(), adjoint=False, method="rk4", options=dict(step_size=1.0))
        ref32 = oracle_cde.cdeint(oracle_interp.CubicPath(coeffs), func, z0, t_out, adjoint=False, method="rk4", options=dict(step_size=1.0))
    dfunc = LinearField(H, C, torch.float32, scale=scale, tanh=True, seed=9).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    with torch.no_grad():
        for variant in ("mfma", "generic"):
            out = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0), variant=variant).cpu().double()
            print(scale, variant, "max abs err vs f64", (out - ref).abs().max().item(), "at t1", (out[:, 1] - ref[:, 1]).abs().max().item(), "max |ref|", ref.abs().max().item())
    print(scale, "cpu f32 oracle vs f64", (ref32.double() - ref).abs().max().item())
# tanh accuracy
