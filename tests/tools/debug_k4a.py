"""Calibration / debugging aid for K4a (not a test): runs the default adaptive call on the GPU, replays EVERY backward
attempt the kernel made through the float64 oracle and prints how the two error ratios compare.

    python tests/tools/debug_k4a.py [case ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import LinearField, TwoLayerField, make_series  # noqa: E402
from oracle import cde as oracle_cde, interp as oracle_interp, odeint as oracle_ode  # noqa: E402
import torchcde_amd as native  # noqa: E402

DEV = "cuda"
CASES = {
    "cubic_tanh_multi_out": dict(B=70, L=10, C=5, H=24, tanh=True, degree=3, t_out=[0., 3.6, 9.], jumps=False,
                                 kw=dict(rtol=1e-5, atol=1e-7), adj={}),
    "linear_jumps": dict(B=130, L=9, C=8, H=32, tanh=False, degree=1, t_out=None, jumps=True,
                         kw=dict(rtol=1e-4, atol=1e-6), adj={}),
    "cubic_seminorm": dict(B=50, L=8, C=6, H=20, tanh=True, degree=3, t_out=None, jumps=False,
                           kw=dict(rtol=1e-4, atol=1e-6), adj=dict(adjoint_options=dict(norm="seminorm"))),
    "cubic_identity_loose": dict(B=64, L=12, C=8, H=32, tanh=False, degree=3, t_out=[0., 11.], jumps=False,
                                 kw=dict(rtol=1e-3, atol=1e-5), adj={}),
    # two-layer fields (K4am): width, final tanh
    "mlp_cubic": dict(B=70, L=10, C=8, H=32, tanh=True, degree=3, t_out=None, jumps=False, kw=dict(rtol=1e-4, atol=1e-6),
                      adj={}, width=128),
    "mlp_linear_jumps_multi": dict(B=150, L=9, C=4, H=16, tanh=True, degree=1, t_out=[0., 3.5, 8.], jumps=True,
                                   kw=dict(rtol=1e-4, atol=1e-6), adj={}, width=64),
    "mlp_16x16_seminorm": dict(B=40, L=8, C=14, H=8, tanh=True, degree=1, t_out=None, jumps=False,
                               kw=dict(rtol=1e-4, atol=1e-6), adj=dict(adjoint_options=dict(norm="seminorm")), width=128),
    "mlp_identity_many_tiles": dict(B=2100, L=7, C=3, H=8, tanh=False, degree=3, t_out=None, jumps=False,
                                    kw=dict(rtol=1e-3, atol=1e-5), adj={}, width=100),
}


def make_field(cfg, dtype):
    if "width" in cfg:
        return TwoLayerField(cfg["H"], cfg["C"], cfg["width"], dtype, seed=3, final_tanh=cfg["tanh"])
    return LinearField(cfg["H"], cfg["C"], dtype, scale=0.3, tanh=cfg["tanh"], seed=7)


def params_of(f):
    return [p for _, p in f.named_parameters()]


def run(name):
    import importlib
    front = importlib.import_module("torchcde_amd.cdeint")
    cfg = CASES[name]
    B, L, C, H = cfg["B"], cfg["L"], cfg["C"], cfg["H"]
    x = make_series(B, L, C, seed=len(name))
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(len(name)))
    t_out = None if cfg["t_out"] is None else torch.tensor(cfg["t_out"])
    n_t = 2 if t_out is None else t_out.numel()
    lw = torch.rand(B, n_t, H, generator=torch.Generator().manual_seed(3)) + 0.5
    func = make_field(cfg, torch.float32).to(DEV)
    X = (native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV))) if cfg["degree"] == 3
         else native.LinearInterpolation(native.linear_interpolation_coeffs(x.to(DEV))))
    zd = z0.to(DEV).requires_grad_(True)
    times = X.interval if t_out is None else t_out.to(DEV)
    opts = dict(options=dict(jump_t=X.grid_points)) if cfg["jumps"] else {}
    front.record_dopri5_steps = True
    try:
        out = native.cdeint(X, func, zd, times, **opts, **cfg["adj"], **cfg["kw"])
        print("   grad_fn:", type(out.grad_fn).__name__)
        fwd = dict(front.last_dopri5_stats)
        (out * lw.to(DEV)).sum().backward()
        bwd = dict(front.last_dopri5_adjoint_stats)
    finally:
        front.record_dopri5_steps = False
    print("==", name, "forward accepted/rejected", fwd["n_accept"], fwd["n_reject"], "backward", bwd["n_accept"],
          bwd["n_reject"], "launches", bwd["launches"])

    # float64 oracle: the kernel's forward steps, then every backward attempt
    solvers = []
    original = oracle_ode._Dopri5.integrate

    def integrate(self, t):
        res = original(self, t)
        solvers.append(self)
        return res

    oracle_ode._Dopri5.integrate = integrate
    try:
        f64 = make_field(cfg, torch.float64)
        Xo = (oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x.double())) if cfg["degree"] == 3
              else oracle_interp.LinearPath(x.double()))
        zo = z0.double().requires_grad_(True)
        to = Xo.interval if t_out is None else t_out.double()
        adj_opts = dict(cfg["adj"].get("adjoint_options", {}))
        adj_opts["replay_attempts"] = [a.clone() for a in bwd["attempts"]]
        if cfg["jumps"]:
            adj_opts["jump_t"] = Xo.grid_points
        fopts = dict(replay_steps=fwd["steps"])
        if cfg["jumps"]:
            fopts["jump_t"] = Xo.grid_points
        ref = oracle_cde.cdeint(Xo, f64, zo, to, adjoint=True, method="dopri5", options=fopts, adjoint_options=adj_opts,
                                **cfg["kw"])
        (ref * lw.double()).sum().backward()
    finally:
        oracle_ode._Dopri5.integrate = original
    for interval, (attempts, solver) in enumerate(zip(bwd["attempts"], solvers[1:])):
        mine = attempts[:, 4]
        theirs = torch.tensor(solver.ratios, dtype=torch.float64)
        accepted = attempts[:, 3] != 0
        rel = (mine - theirs).abs() / theirs.clamp_min(1e-3)
        wrong = ((theirs > 1) == accepted)
        print("  interval %d: %d attempts, first dt kernel %.6g oracle %.6g; ratio rel.dev median %.3g max %.3g; decisions "
              "contradicting the oracle's ratio: %d (|ratio-1| of those: %s)"
              % (interval, len(mine), float(attempts[0, 1] - attempts[0, 0]), float(solver.first_dt), rel.median(), rel.max(),
                 int(wrong.sum()), [round(float(v), 4) for v in (theirs[wrong] - 1).abs()[:8]]))
        worst = rel.argsort(descending=True)[:5]
        for j in worst:
            print("     attempt %d  t0 %.6f dt %.3g  kernel %.5g oracle %.5g accepted %d" %
                  (j, attempts[j, 0], attempts[j, 1] - attempts[j, 0], mine[j], theirs[j], int(attempts[j, 3])))

    def show(label, got, want):
        got, want = got.detach().double().cpu(), want.detach().double()
        print("  %-8s max abs err %.3g  (scale %.3g)" % (label, (got - want).abs().max(), want.abs().max()))
    show("z", out, ref)
    show("dz0", zd.grad, zo.grad)
    for (name_p, p_gpu), p_ref in zip(func.named_parameters(), params_of(f64)):
        show(name_p, p_gpu.grad, p_ref.grad)


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        run(name)
