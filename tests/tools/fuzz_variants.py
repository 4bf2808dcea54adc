"""Randomised cross-check of the kernel families (not part of pytest; run on a GPU box):

    python tests/tools/fuzz_variants.py --cases 150 --seed 1

Every case draws a shape (batch incl. 1 / 15 / 16 / 17, ragged tiles; hidden and control channels across the tile
boundaries 8 / 16 / 32 / 64), a field (affine, tanh, two-layer), a control (cubic / linear, regular or irregular
knots, optional extra batch dimension), output times off the knots and a solver setting, runs it once under
`variant="auto"` (MFMA / wide / split / two-layer kernels, whichever AUTO picks) and once under `variant="generic"`
(VALU kernels / step-wise path) and compares trajectories and every gradient.  Prints the failing configurations."""
import argparse, os, random, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde
from helpers import LinearField


class TwoLayer(torch.nn.Module):
    def __init__(self, H, C, width, final_tanh, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.H, self.C, self.final_tanh = H, C, final_tanh
        self.linear1, self.linear2 = torch.nn.Linear(H, width), torch.nn.Linear(width, H * C)

    def forward(self, t, z):
        y = self.linear2(self.linear1(z).relu())
        if self.final_tanh:
            y = y.tanh()
        return y.view(*z.shape[:-1], self.H, self.C)


def draw(rng):
    kind = rng.choice(["affine", "affine", "tanh", "two_layer"])
    if kind == "two_layer":
        # (round 6: 17..32 hidden units x 9..16 channels -- the upper half of the output layer read from L2)
        H, C = rng.choice([(8, 3), (16, 14), (32, 8), (5, 2), (16, 16), (32, 4), (12, 9), (32, 14), (24, 16), (20, 9), (32, 16),
                           (17, 12)])
    else:
        H = rng.choice([1, 3, 8, 16, 24, 31, 32, 33, 48, 64, 70])
        C = rng.choice([1, 2, 4, 7, 8, 9, 14, 16, 17]) if H <= 32 else rng.choice([1, 3, 8, 9])
    return dict(kind=kind, H=H, C=C, width=rng.choice([8, 32, 100, 128]), final_tanh=rng.random() < 0.6,
                B=rng.choice([1, 2, 15, 16, 17, 33, 100, 257]), L=rng.choice([2, 3, 5, 12, 30]),
                degree=rng.choice([1, 3]), irregular=rng.random() < 0.5, extra_dim=rng.random() < 0.2,
                mode=rng.choice(["rk4", "rk4", "dopri5_forward", "default_call", "midpoint", "euler", "rk4_backprop",
                                 "rk4_backprop_control", "rk4_control", "default_call_control", "default_call_control"]),
                # round 6 (the *_control modes): adjoint_params = parameters + (coeffs[, t]); the coefficients a leaf or fitted
                # inside the graph with the same knot tensor (the reference's grad-paths scenario: the fit's chain)
                knots=rng.random() < 0.5, fit_chain=rng.random() < 0.4, mixed_norm=rng.random() < 0.4,
                step=rng.choice([1.0, 0.5, 0.37]), n_out=rng.choice([2, 3, 5]), seed=rng.randrange(10 ** 6))


def make_field(cfg, dev):
    if cfg["kind"] == "two_layer":
        return TwoLayer(cfg["H"], cfg["C"], cfg["width"], cfg["final_tanh"], cfg["seed"]).to(dev)
    return LinearField(cfg["H"], cfg["C"], scale=0.3, tanh=cfg["kind"] == "tanh", seed=cfg["seed"]).to(dev)


def run(cfg, variant, dev):
    gen = torch.Generator().manual_seed(cfg["seed"])
    B, L, C, H = cfg["B"], cfg["L"], cfg["C"], cfg["H"]
    lead = (2, B) if cfg["extra_dim"] else (B,)
    x = (torch.randn(*lead, L, C, generator=gen) * 0.3).cumsum(-2).to(dev)
    if cfg["mode"] == "rk4_backprop_control":         # the data require a gradient: fit -> control -> solve (adjoint=False) -> loss
        x.requires_grad_(True)
    t = ((torch.rand(L, generator=gen) + 0.4).cumsum(0) if cfg["irregular"] else torch.arange(L, dtype=torch.float32)).to(dev)
    control_mode = cfg["mode"] in ("rk4_control", "default_call_control")
    leaves = []
    if control_mode:
        if cfg["knots"]:
            t.requires_grad_(True)
            leaves.append(t)
        if cfg["fit_chain"]:
            x.requires_grad_(True)
            leaves.append(x)
    fit = cde.hermite_cubic_coefficients_with_backward_differences if cfg["degree"] == 3 else cde.linear_interpolation_coeffs
    coeffs = fit(x, t) if (not control_mode or cfg["fit_chain"]) else fit(x, t.detach()).detach().requires_grad_(True)
    if control_mode and not cfg["fit_chain"]:
        leaves.append(coeffs)
    X = (cde.CubicSpline if cfg["degree"] == 3 else cde.LinearInterpolation)(coeffs, t)
    lo, hi = t[0].item(), t[-1].item()
    inner = torch.sort(torch.rand(cfg["n_out"] - 2, generator=gen) * (hi - lo) + lo).values
    t_out = torch.cat([torch.tensor([lo]), inner, torch.tensor([hi])]).to(dev)
    z0 = torch.randn(*lead, H, generator=gen).to(dev).requires_grad_(cfg["mode"] != "dopri5_forward")
    w = (torch.rand(*lead, cfg["n_out"], H, generator=gen) + 0.5).to(dev)
    func = make_field(cfg, dev)
    spacing = (hi - lo) / (L - 1)
    if cfg["mode"] == "rk4":
        out = cde.cdeint(X, func, z0, t_out, method="rk4", options=dict(step_size=cfg["step"] * spacing), variant=variant)
    elif cfg["mode"] in ("midpoint", "euler"):        # round 5: K2 / K3p with two stages / one per step (affine field on the tiles)
        out = cde.cdeint(X, func, z0, t_out, method=cfg["mode"], options=dict(step_size=cfg["step"] * spacing), variant=variant)
    elif cfg["mode"] in ("rk4_backprop", "rk4_backprop_control"):   # round 5: adjoint=False, reverse mode through the steps (K3d) vs autograd
        out = cde.cdeint(X, func, z0, t_out, method="rk4", adjoint=False, options=dict(step_size=cfg["step"] * spacing),
                         variant=variant)
    elif cfg["mode"] == "rk4_control":
        out = cde.cdeint(X, func, z0, t_out, method="rk4", options=dict(step_size=cfg["step"] * spacing), variant=variant,
                         adjoint_params=tuple(func.parameters()) + ((coeffs, t) if cfg["knots"] else (coeffs,)))
    elif cfg["mode"] == "default_call_control":
        # under "seminorm" the coefficient / knot blocks are integrated but not error-controlled, and their integrands jump at
        # the knots (row e is read inside interval e only): two different step sequences -- this comparison -- then differ by
        # percents unless the steps stop at the knots (seed 311 of round 6: three such cases, each within 1e-5 of the float64
        # oracle replaying the fused run's own steps: profiles/r06_fuzz_seed311.log)
        opts = dict(jump_t=X.grid_points.detach()) if (cfg["degree"] == 1 or not cfg["mixed_norm"]) else {}
        adj = dict(opts) if cfg["mixed_norm"] else dict(norm="seminorm", **opts)
        out = cde.cdeint(X, func, z0, t_out, rtol=1e-6, atol=1e-8, options=opts, adjoint_options=adj, variant=variant,
                         adjoint_params=tuple(func.parameters()) + ((coeffs, t) if cfg["knots"] else (coeffs,)))
    elif cfg["mode"] == "dopri5_forward":
        opts = dict(jump_t=X.grid_points) if cfg["degree"] == 1 else {}
        with torch.no_grad():
            out = cde.cdeint(X, func, z0, t_out, method="dopri5", rtol=1e-6, atol=1e-8, options=opts, variant=variant)
        return [out]
    else:
        opts = dict(jump_t=X.grid_points) if cfg["degree"] == 1 else {}
        out = cde.cdeint(X, func, z0, t_out, rtol=1e-6, atol=1e-8, options=opts,
                         adjoint_options=dict(norm="seminorm", **opts), variant=variant)
    (out * w).sum().backward()
    return ([out.detach(), z0.grad] + [p.grad for p in func.parameters()] + ([x.grad] if x.requires_grad and not control_mode else [])
            + [leaf.grad for leaf in leaves])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = random.Random(a.seed)
    bad = 0
    for i in range(a.cases):
        cfg = draw(rng)
        try:
            got, want = run(cfg, "auto", dev), run(cfg, "generic", dev)
            tol = 2e-3 if cfg["mode"] in ("rk4", "midpoint", "euler", "rk4_backprop", "rk4_backprop_control", "rk4_control") else 2e-2
            for k, (g, r) in enumerate(zip(got, want)):
                scale = max(r.abs().max().item(), 1e-3)
                err = (g - r).abs().max().item()
                if not (err <= tol * scale) or not torch.isfinite(g).all():
                    bad += 1
                    print("MISMATCH case %d tensor %d: err %.3e scale %.3e  %s" % (i, k, err, scale, cfg), flush=True)
                    break
        except Exception as exc:                                   # noqa: BLE001 -- report and go on
            bad += 1
            print("ERROR case %d: %r  %s" % (i, exc, cfg), flush=True)
            traceback.print_exc(limit=3)
    print("fuzz: %d cases, %d bad" % (a.cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
