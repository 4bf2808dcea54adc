import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as native
from oracle import interp as oi, cde as oc
from test_gpu_parity import _TwoLayerField
DEV = "cuda:0"
gen = torch.Generator().manual_seed(14)
B, L, H, C = 1, 2, 9, 13
x = torch.randn(B, L, C, generator=gen)
coeffs = oi.hermite_bdiff_coeffs(x)
z0 = torch.randn(B, H, generator=gen)
f = _TwoLayerField(H, C, 48, seed=3).to(DEV); f64 = _TwoLayerField(H, C, 48, torch.float64, seed=3)
X, Xo = native.CubicSpline(coeffs.to(DEV)), oi.CubicPath(coeffs.double())
t_out = torch.tensor([0., 1.])
ref = oc.cdeint(Xo, f64, z0.double(), t_out.double(), adjoint=False, method="rk4", options=dict(step_size=0.5))
for rep in range(3):
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, f, z, t_out.to(DEV), method="rk4", options=dict(step_size=0.5))
    print(rep, type(out.grad_fn).__name__, "err", float((out.detach().double().cpu() - ref.detach()).abs().max()), "coeffs", coeffs.flatten()[:6].tolist())
with torch.no_grad():
    out = native.cdeint(X, f, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=0.5))
print("nograd err", float((out.double().cpu() - ref.detach()).abs().max()))
