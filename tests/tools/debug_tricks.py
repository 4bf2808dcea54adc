import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as native
from oracle import interp as oi, cde as oc
dtype = torch.float64
class F(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.variable = torch.nn.Parameter(torch.rand(1, 1, 3, generator=torch.Generator().manual_seed(5), dtype=dtype))
    def forward(s, t, z): return z.sigmoid().unsqueeze(-1) + s.variable
gen = torch.Generator().manual_seed(17)
path0 = torch.rand(1, 10, 3, generator=gen, dtype=dtype); z00 = torch.rand(1, 3, generator=gen, dtype=dtype)
def run(interp, spline, solve, dev, opts, which):
    t = torch.linspace(0, 9, 10, dtype=dtype, device=dev).requires_grad_("t" in which)
    path = path0.to(dev).clone().requires_grad_("path" in which)
    X = spline(interp(path, t), t)
    z0 = z00.to(dev).clone().requires_grad_(True)
    f = F().to(dev)
    t_ = torch.tensor([0., 9.], dtype=dtype, device=dev, requires_grad="t_" in which)
    z = solve(X, f, z0, t_, adjoint=False, method="rk4", **opts)
    z[:, 1].sum().backward()
    return z.detach().cpu(), [None if g is None else g.detach().cpu() for g in (t.grad, path.grad, z0.grad, f.variable.grad, t_.grad)]
for opts in (dict(), dict(options=dict(step_size=0.5))):
    for which in (("t",), ("path",), ("t_",), ("t", "path", "t_")):
        zn, gn = run(native.natural_cubic_coeffs, native.CubicSpline, native.cdeint, "cuda:0", opts, which)
        zo, go = run(oi.natural_cubic_coeffs, oi.CubicPath, oc.cdeint, "cpu", opts, which)
        errs = [None if a is None else float(((a - b).abs().max() / (b.abs().max() + 1e-300))) for a, b in zip(gn, go)]
        if which == ("t", "path", "t_"): print("t grads native", gn[0], "oracle", go[0], "z", zn, zo)
        print(opts, which, "z err %.2e |z| %.2e" % (float((zn - zo).abs().max()), float(zo.abs().max())), "grad rel errs (t,path,z0,var,t_):", errs)
