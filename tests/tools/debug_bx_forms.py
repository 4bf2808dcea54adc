"""Debugging aid (not a test): the four reverse-sweep kernels of the affine field side by side -- K3 / K3j (exact f32),
K3b / K3bj (variant "bf16x3") -- at a given batch: deviations from K3 and, for dL/dz0, from the float64 oracle.
    python tests/tools/debug_bx_forms.py [batch] [length]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402
from oracle import cde as oracle_cde, interp as oracle_interp  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
C, H = 8, 32
x = make_series(B, L, C, seed=0)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0))
X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV)))
res = {}
for name, variant, form in (("k3j", "mfma", "j"), ("k3", "mfma", "product"), ("k3bj", "bf16x3", "j"), ("k3b", "bf16x3", "product")):
    os.environ["CDE_K3_FORM"] = form
    f = LinearField(H, C, scale=0.25, seed=0).to(DEV)
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, f, z, X.interval, method="rk4", options=dict(step_size=1.0), variant=variant)
    out[:, -1].sum().backward()
    res[name] = tuple(t.detach().cpu().double() for t in (out, z.grad, f.linear.weight.grad, f.linear.bias.grad))
    print(name, "finite:", [bool(torch.isfinite(t).all()) for t in res[name]],
          "series with non-finite dz0:", int((~torch.isfinite(res[name][1]).all(dim=1)).sum()))
n = min(B, 2048)
f64 = LinearField(H, C, torch.float64, scale=0.25, seed=0)
Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x[:n].double()))
zo = z0[:n].double().requires_grad_(True)
o = oracle_cde.cdeint(Xo, f64, zo, Xo.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
o[:, -1].sum().backward()
for name in res:
    print(name, "dz0 vs oracle: max err / scale %.3g" % float((res[name][1][:n] - zo.grad).abs().max() / zo.grad.abs().max()))
for name in ("k3j", "k3bj", "k3b"):
    print(name, "vs k3 (z, dz0, dW, db):", ["%.3g" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[name], res["k3"])])
