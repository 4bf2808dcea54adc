"""bf16x3 variant vs the f32 MFMA kernels vs the float64 oracle: accuracy and time (not a test)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde
from helpers import LinearField, make_series
from oracle import cde as oracle_cde, interp as oracle_interp
DEV = "cuda"
# ---- small: against the float64 oracle
for (B, L, C, H, degree) in ((75, 12, 8, 32, 3), (40, 9, 5, 20, 1)):
    x = make_series(B, L, C, seed=5)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(5))
    t_out = torch.tensor([0., 4.5, float(L - 1)])
    lw = torch.rand(B, 3, H, generator=torch.Generator().manual_seed(6)) + 0.5
    f64 = LinearField(H, C, torch.float64, scale=0.3, seed=3)
    Xo = (oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x.double())) if degree == 3 else oracle_interp.LinearPath(x.double()))
    zo = z0.double().requires_grad_(True)
    ref = oracle_cde.cdeint(Xo, f64, zo, t_out.double(), adjoint=True, method="rk4", options=dict(step_size=1.0))
    (ref * lw.double()).sum().backward()
    for variant in ("mfma", "bf16x3"):
        f = LinearField(H, C, scale=0.3, seed=3).to(DEV)
        X = (cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x.to(DEV))) if degree == 3
             else cde.LinearInterpolation(cde.linear_interpolation_coeffs(x.to(DEV))))
        z = z0.to(DEV).requires_grad_(True)
        out = cde.cdeint(X, f, z, t_out.to(DEV), method="rk4", options=dict(step_size=1.0), variant=variant)
        (out * lw.to(DEV)).sum().backward()
        def err(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
        print("B=%d H=%d C=%d deg=%d %-7s rel.err  z %.2e  dz0 %.2e  dW %.2e  db %.2e" % (
            B, H, C, degree, variant, err(out.detach(), ref.detach()), err(z.grad, zo.grad),
            err(f.linear.weight.grad, f64.linear.weight.grad), err(f.linear.bias.grad, f64.linear.bias.grad)))
# ---- headline size: time
B, L, C, H = 32768, 128, 8, 32
x = make_series(B, L, C, seed=0).to(DEV)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
f = LinearField(H, C, scale=0.25, seed=0).to(DEV)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(DEV)
res = {}
for variant in ("mfma", "bf16x3"):
    def fwd():
        with torch.no_grad():
            return cde.cdeint(X, f, z0, X.interval, method="rk4", options=dict(step_size=1.0), variant=variant)
    def both():
        z = z0.detach().requires_grad_(True)
        for p in f.parameters(): p.grad = None
        o = cde.cdeint(X, f, z, X.interval, method="rk4", options=dict(step_size=1.0), variant=variant)
        o[:, -1].sum().backward()
        return o.detach(), z.grad, f.linear.weight.grad.clone()
    for fn, name in ((fwd, "forward"), (both, "fwd+adjoint")):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = fn()
        torch.cuda.synchronize()
        print("%-7s %-12s %.3f ms" % (variant, name, (time.perf_counter() - t0) / 10 * 1e3))
    res[variant] = both()
for a, b, name in zip(res["bf16x3"], res["mfma"], ("z", "dz0", "dW")):
    print("bf16x3 vs mfma at size: %s max rel diff %.2e" % (name, float((a - b).abs().max() / b.abs().max())))
