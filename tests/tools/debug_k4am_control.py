"""Debug aid (round 6): which block of torchdiffeq's mixed adjoint norm decides an attempt -- the float64 oracle replays the
attempts of K4am with control gradients under a logging norm; attempts whose ratio differs from the kernel's are listed with
the oracle's per-block values (|vjp_t|, rms y, rms a, rms of every adjoint_params entry).
    python tests/tools/debug_k4am_control.py [B] [knots 0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native  # noqa: E402
from gpu_common import oracle_cde, oracle_interp, _TwoLayerField, make_series, DEV, _front  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4400
with_knots = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
L, C, H, width, kw = 6, 8, 32, 128, dict(rtol=1e-4, atol=1e-6)
front = _front()
x = make_series(B, L, C, seed=25)
gaps = torch.rand(L - 1, generator=torch.Generator().manual_seed(6)) + 0.5
knots0 = torch.cat([torch.zeros(1), gaps.cumsum(0)]) * ((L - 1) / gaps.sum()) if with_knots else None
base = oracle_interp.hermite_bdiff_coeffs(x, knots0)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(25))
t_out = torch.tensor([0., 5.])
lw = torch.rand(B, 2, H, generator=torch.Generator().manual_seed(3)) + 0.5
func = _TwoLayerField(H, C, width, seed=3).to(DEV)
coeffs = base.to(DEV).requires_grad_(True)
kd = knots0.to(DEV).requires_grad_(True) if with_knots else None
X = native.CubicSpline(coeffs, kd)
zd = z0.to(DEV).requires_grad_(True)
front.record_dopri5_steps = True
out = native.cdeint(X, func, zd, t_out.to(DEV), adjoint_params=tuple(func.parameters()) + ((coeffs, kd) if with_knots else (coeffs,)), **kw)
print("dispatch:", front.last_dispatch()[0])
fwd = dict(front.last_dopri5_stats)
(out * lw.to(DEV)).sum().backward()
bwd = dict(front.last_dopri5_adjoint_stats)
front.record_dopri5_steps = False

f64 = _TwoLayerField(H, C, width, torch.float64, seed=3)
c64 = base.double().clone().requires_grad_(True)
ko = knots0.double().requires_grad_(True) if with_knots else None
Xo = oracle_interp.CubicPath(c64, ko)
zo = z0.double().requires_grad_(True)
log = []


def rms(v):
    return float(v.pow(2).mean().sqrt())


def norm(parts):
    tt, yy, aa, *pp = parts
    vals = [float(tt.abs()), rms(yy), rms(aa)] + [rms(p) for p in pp]
    log.append(vals)
    return torch.tensor(max(vals), dtype=torch.float64)


ref = oracle_cde.cdeint(Xo, f64, zo, t_out.double(), adjoint=True, method="dopri5", options=dict(replay_steps=fwd["steps"]),
                        adjoint_options=dict(replay_attempts=[a.clone() for a in bwd["attempts"]], norm=norm),
                        adjoint_params=tuple(f64.parameters()) + ((c64, ko) if with_knots else (c64,)), **kw)
(ref * lw.double()).sum().backward()
attempts = bwd["attempts"][0]
names = ["vjp_t", "y", "a", "W1", "b1", "W2", "b2", "coeffs"] + (["knots"] if with_knots else [])
# the logging norm is also called by the initial-step selection (3 calls) before the attempts
rows = log[-len(attempts):]
bad = 0
for i, (row, vals) in enumerate(zip(attempts, rows)):
    mine, theirs = float(row[4]), max(vals)
    if abs(mine - theirs) > 0.02 * theirs + 0.01:
        bad += 1
        top = sorted(zip(vals, names), reverse=True)[:3]
        print("attempt %4d  t0 %.4f dt %.5f acc %d  kernel %.4f oracle %.4f   oracle blocks: %s" % (
            i, row[0], row[1] - row[0], int(row[3]), mine, theirs, ", ".join("%s %.4f" % (n, v) for v, n in top)))
print("%d of %d attempts outside the band" % (bad, len(attempts)))
print("dL/dcoeffs max rel err", float((coeffs.grad.double().cpu() - c64.grad).abs().max() / c64.grad.abs().max()))
if with_knots:
    print("dL/dknots", kd.grad.cpu().tolist(), ko.grad.tolist())
