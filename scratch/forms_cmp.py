import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as native
from test_gpu_parity import _TwoLayerField, make_series
DEV = "cuda"
B, L, C, H, width = int(sys.argv[1]), 17, 8, 32, 128
x = make_series(B, L, C, seed=11)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(11))
func = _TwoLayerField(H, C, width, seed=5).to(DEV)
X = native.LinearInterpolation(native.linear_interpolation_coeffs(x.to(DEV)))
res = {}
for form in ("default", "one_wave"):
    for k in ("CDE_K4AM_NO_SPLIT", "CDE_K4M_NO_SPLIT"):
        os.environ.pop(k, None)
        if form == "one_wave": os.environ[k] = "1"
    zd = z0.to(DEV).requires_grad_(True)
    func.zero_grad()
    out = native.cdeint(X, func, zd, X.interval, adjoint_options=dict(norm="seminorm"), rtol=1e-4, atol=1e-6)
    front = sys.modules["torchcde_amd.cdeint"]
    print(form, "fwd", {k: v for k, v in front.last_dopri5_stats.items() if k.startswith("n_")})
    out[:, -1].sum().backward()
    print(form, "bwd", {k: v for k, v in front.last_dopri5_adjoint_stats.items() if k.startswith("n_")})
    res[form] = (out.detach().cpu(), zd.grad.cpu(), [p.grad.cpu().clone() for p in func.parameters()])
a, b = res["default"], res["one_wave"]
print("out diff", (a[0]-b[0]).abs().max().item())
d = (a[1]-b[1]).abs()
print("gz diff quantiles", torch.quantile(d.max(1).values, torch.tensor([0.5,0.9,0.99,1.0])).tolist())
print("gz diff max", d.max().item(), "of", b[1].abs().max().item(), "rows > 1e-3:", (d.max(1).values > 1e-3).sum().item())
for p, q in zip(a[2], b[2]):
    print("param diff", (p-q).abs().max().item(), "of", q.abs().max().item())
