"""A flagged `default_call_control` case of tests/tools/fuzz_variants.py (fused vs step-wise: two DIFFERENT adaptive step
sequences) against the float64 oracle replaying the FUSED run's own steps -- which separates "the kernels integrate something
else" from "two step sequences integrate a discontinuous integrand differently" (under "seminorm" the coefficient / knot blocks
are not error-controlled, and their integrands jump at the knots).
    python tests/tools/check_fuzz_case_against_oracle.py "<the cfg dict the fuzz tool printed>" """
import ast
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import torchcde_amd as cde  # noqa: E402
import fuzz_variants as F  # noqa: E402
from gpu_common import oracle_cde, oracle_interp, LinearField, _front  # noqa: E402

cfg = ast.literal_eval(sys.argv[1])
assert cfg["mode"] == "default_call_control" and cfg["kind"] in ("affine", "tanh") and not cfg["extra_dim"]
dev = torch.device("cuda", 0)
front = _front()
front.record_dopri5_steps = True
got = F.run(cfg, "auto", dev)
print("dispatch:", front.last_dispatch()[0])
fwd, bwd = dict(front.last_dopri5_stats), dict(front.last_dopri5_adjoint_stats)
front.record_dopri5_steps = False

# the same inputs in float64 on the CPU (fuzz_variants.run's draws, in its order)
gen = torch.Generator().manual_seed(cfg["seed"])
B, L, C, H = cfg["B"], cfg["L"], cfg["C"], cfg["H"]
x = (torch.randn(B, L, C, generator=gen) * 0.3).cumsum(-2).double()
t = ((torch.rand(L, generator=gen) + 0.4).cumsum(0) if cfg["irregular"] else torch.arange(L, dtype=torch.float32)).double()
leaves = []
if cfg["knots"]:
    t.requires_grad_(True)
    leaves.append(t)
if cfg["fit_chain"]:
    x.requires_grad_(True)
    leaves.append(x)
fit = oracle_interp.hermite_bdiff_coeffs if cfg["degree"] == 3 else (lambda xx, tt: xx)
coeffs = fit(x, t) if cfg["fit_chain"] else fit(x, t.detach()).detach().requires_grad_(True)
if not cfg["fit_chain"]:
    leaves.append(coeffs)
X = (oracle_interp.CubicPath if cfg["degree"] == 3 else oracle_interp.LinearPath)(coeffs, t)
lo, hi = float(t[0]), float(t[-1])
inner = torch.sort(torch.rand(cfg["n_out"] - 2, generator=gen) * (hi - lo) + lo).values
t_out = torch.cat([torch.tensor([lo]), inner, torch.tensor([hi])]).float().double()
z0 = torch.randn(B, H, generator=gen).double().requires_grad_(True)
w = (torch.rand(B, cfg["n_out"], H, generator=gen) + 0.5).double()
func = LinearField(H, C, torch.float64, scale=0.3, tanh=cfg["kind"] == "tanh", seed=cfg["seed"])
opts = dict(replay_steps=fwd["steps"])
adj = dict(replay_attempts=[a.clone() for a in bwd["attempts"]])
if not cfg["mixed_norm"]:
    adj["norm"] = "seminorm"
out = oracle_cde.cdeint(X, func, z0, t_out, adjoint=True, method="dopri5", rtol=1e-6, atol=1e-8, options=opts, adjoint_options=adj,
                        adjoint_params=tuple(func.parameters()) + ((coeffs, t) if cfg["knots"] else (coeffs,)))
(out * w).sum().backward()
want = [out.detach(), z0.grad] + [p.grad for p in func.parameters()] + [leaf.grad for leaf in leaves]
names = ["out", "z0", "W", "b"] + ["t" if leaf is t else "x" if leaf is x else "coeffs" for leaf in leaves]
for n, g, r in zip(names, got, want):
    g = g.double().cpu()
    print("%-7s fused vs oracle replaying the fused steps: max err %.3e of scale %.3e" % (n, (g - r).abs().max(), r.abs().max()))
