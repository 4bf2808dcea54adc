"""Debug aid (round 6): dL/dz0 of the 32-unit x 14-channel two-layer field under rk4, fused kernels vs the float64 oracle,
per series -- relu-kink noise shows as a few outlying series, a systematic error as all of them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native  # noqa: E402
from gpu_common import oracle_cde, oracle_interp, _TwoLayerField, make_series, DEV  # noqa: E402

H, C, width, degree, final_tanh = 32, 14, 128, 3, True
B, L = 203, 24
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 71
if len(sys.argv) > 2:
    H, C, width = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = make_series(B, L, C, torch.float32, seed=seed)
coeffs = oracle_interp.hermite_bdiff_coeffs(x)
gen = torch.Generator().manual_seed(seed + 1)
z0 = torch.randn(B, H, generator=gen)
t_out = torch.tensor([0., 7.5, 23.])
lw = torch.rand(B, 3, H, generator=gen) + 0.5
res = {}
for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
    f = _TwoLayerField(H, C, width, dtype, seed=5, final_tanh=final_tanh)
    path = oracle_interp.CubicPath(coeffs.to(dtype))
    z = z0.to(dtype).clone().requires_grad_(True)
    out = oracle_cde.cdeint(path, f, z, t_out.to(dtype), adjoint=True, method="rk4", options=dict(step_size=1.0))
    (out * lw.to(dtype)).sum().backward()
    res[name] = (z.grad.double(), [p.grad.double() for p in f.parameters()])
for name, kw in (("fused", {}), ("stepwise", dict(variant="generic"))):
    f = _TwoLayerField(H, C, width, seed=5, final_tanh=final_tanh).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, f, z, t_out.to(DEV), method="rk4", options=dict(step_size=1.0), **kw)
    (out * lw.to(DEV)).sum().backward()
    res[name] = (z.grad.double().cpu(), [p.grad.double().cpu() for p in f.parameters()])
want = res["f64"][0]
scale = want.abs().max().item()
print("seed %d: max |dL/dz0| = %.4g" % (seed, scale))
for name in ("f32", "fused", "stepwise"):
    err = (res[name][0] - want).abs().max(dim=1).values / scale
    worst = torch.topk(err, 5)
    print("%-9s per-series max error / scale: median %.3g, 90%% %.3g, worst five %s at series %s" % (
        name, err.median().item(), err.quantile(0.9).item(), ["%.3g" % v for v in worst.values.tolist()], worst.indices.tolist()))
    for i, (g, w) in enumerate(zip(res[name][1], res["f64"][1])):
        print("          param %d: max err / max = %.3g" % (i, ((g - w).abs().max() / w.abs().max()).item()))
