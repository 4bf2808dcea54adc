import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as native
from oracle import interp as oi, cde as oc
from test_gpu_parity import _TwoLayerField
DEV = "cuda:0"
gen = torch.Generator().manual_seed(0)
for (B, L, H, C, step) in ((1, 2, 9, 13, 0.5), (19, 2, 9, 13, 0.5), (2, 4, 9, 13, 0.5), (1, 2, 12, 5, 0.5), (19, 2, 12, 5, 0.5), (2, 4, 12, 5, 0.5), (1, 2, 12, 5, 1.0), (16, 2, 32, 8, 0.5), (16, 3, 32, 8, 0.5), (16, 3, 12, 5, 1.0)):
    x = torch.randn(B, L, C, generator=gen)
    coeffs = oi.hermite_bdiff_coeffs(x)
    z0 = torch.randn(B, H, generator=gen)
    f = _TwoLayerField(H, C, 48, seed=3).to(DEV); f64 = _TwoLayerField(H, C, 48, torch.float64, seed=3)
    X, Xo = native.CubicSpline(coeffs.to(DEV)), oi.CubicPath(coeffs.double())
    t_out = torch.tensor([0., float(L - 1)])
    with torch.no_grad():
        out = native.cdeint(X, f, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=step))
        ref = oc.cdeint(Xo, f64, z0.double(), t_out.double(), adjoint=False, method="rk4", options=dict(step_size=step))
    z = z0.to(DEV).requires_grad_(True)
    outg = native.cdeint(X, f, z, t_out.to(DEV), method="rk4", options=dict(step_size=step))
    t1 = torch.tensor([0.]).to(DEV)
    out1 = native.cdeint(X, f, z, t1, method="rk4", options=dict(step_size=step))
    print("   grad path:", type(outg.grad_fn).__name__, "err", float((outg.detach().double().cpu() - ref).abs().max()), "single-time", tuple(out1.shape), float((out1[:, 0] - z).abs().max()))
    print((B, L, H, C, step), "max err", float((out.double().cpu() - ref).abs().max()), "moved", float((out[:, 1] - out[:, 0]).abs().max()))
