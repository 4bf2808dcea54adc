"""Debug aid (round 6): the reference's grad-paths scenario (test/test_tricks.py:21-49) through K4a with the coefficient and
knot blocks -- which block of torchdiffeq's mixed norm decides the attempts whose error ratio differs from the float64 oracle's.
    python tests/tools/debug_k4a_control.py [fit natural|hermite] [knots 0|1] [times 0|1] [data rand|series] [H] [C]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native  # noqa: E402
from gpu_common import oracle_cde, oracle_interp, LinearField, make_series, DEV, _front  # noqa: E402

fit = sys.argv[1] if len(sys.argv) > 1 else "natural"
with_knots = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
with_times = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
data = sys.argv[4] if len(sys.argv) > 4 else "rand"
H = int(sys.argv[5]) if len(sys.argv) > 5 else 3
C = int(sys.argv[6]) if len(sys.argv) > 6 else 3
print("fit", fit, "knots", with_knots, "times", with_times, "data", data, "H", H, "C", C)
front = _front()
B, L, kw = 24, 10, dict(rtol=1e-4, atol=1e-6)
gen = torch.Generator().manual_seed(17)
path0 = torch.rand(B, L, C, generator=gen) if data == "rand" else make_series(B, L, C, seed=17)
z00 = torch.rand(B, H, generator=gen)
gaps = torch.rand(L - 1, generator=gen) + 0.5
t0 = torch.cat([torch.zeros(1), gaps.cumsum(0)]) * ((L - 1) / gaps.sum())

t = t0.to(DEV).requires_grad_(True)
path = path0.to(DEV).requires_grad_(True)
coeffs = (native.natural_cubic_coeffs if fit == "natural" else native.hermite_cubic_coefficients_with_backward_differences)(path, t)
X = native.CubicSpline(coeffs, t)
z0 = z00.to(DEV).requires_grad_(True)
func = LinearField(H, C, scale=0.4, tanh=True, seed=7).to(DEV)
t_ = torch.tensor([0., 4.3, 9.], device=DEV, requires_grad=with_times)
front.record_dopri5_steps = True
z = native.cdeint(X, func, z0, t_, adjoint=True, method="dopri5",
                  adjoint_params=tuple(func.parameters()) + ((coeffs, t) if with_knots else (coeffs,)), **kw)
print("dispatch:", front.last_dispatch()[0])
fwd = dict(front.last_dopri5_stats)
z[:, 1:].sum().backward()
bwd = dict(front.last_dopri5_adjoint_stats)
front.record_dopri5_steps = False

to = t0.double().requires_grad_(True)
po = path0.double().requires_grad_(True)
co = (oracle_interp.natural_cubic_coeffs if fit == "natural" else oracle_interp.hermite_bdiff_coeffs)(po, to)
Xo = oracle_interp.CubicPath(co, to)
zo = z00.double().requires_grad_(True)
f64 = LinearField(H, C, torch.float64, scale=0.4, tanh=True, seed=7)
t_o = torch.tensor([0., 4.3, 9.], dtype=torch.float64, requires_grad=with_times)
log = []


def rms(v):
    return float(v.pow(2).mean().sqrt())


def norm(parts):
    tt, yy, aa, *pp = parts
    vals = [float(tt.abs()), rms(yy), rms(aa)] + [rms(p) for p in pp]
    log.append(vals)
    return torch.tensor(max(vals), dtype=torch.float64)


ref = oracle_cde.cdeint(Xo, f64, zo, t_o, adjoint=True, method="dopri5", options=dict(replay_steps=fwd["steps"]),
                        adjoint_options=dict(replay_attempts=[a.clone() for a in bwd["attempts"]], norm=norm),
                        adjoint_params=tuple(f64.parameters()) + ((co, to) if with_knots else (co,)), **kw)
ref[:, 1:].sum().backward()
names = ["vjp_t", "y", "a", "W", "b", "coeffs"] + (["knots"] if with_knots else [])
n_att = sum(len(a) for a in bwd["attempts"])
n_init = (len(log) - n_att) // len(bwd["attempts"])
pos = 0
for k, attempts in enumerate(bwd["attempts"]):
    pos += n_init
    rows = log[pos:pos + len(attempts)]
    pos += len(attempts)
    bad = 0
    for i, (row, vals) in enumerate(zip(attempts, rows)):
        mine, theirs = float(row[4]), max(vals)
        if abs(mine - theirs) > 0.02 * theirs + 0.01:
            bad += 1
            if bad <= 12:
                print("  interval %d attempt %3d  t0 %.4f dt %.5f acc %d  kernel %.4f oracle %.4f   oracle blocks: %s" % (
                    k, i, row[0], row[1] - row[0], int(row[3]), mine, theirs,
                    ", ".join("%s %.4f" % (n, v) for v, n in sorted(zip(vals, names), reverse=True)[:4])))
    print("interval %d: %d of %d attempts outside the band" % (k, bad, len(attempts)))
pairs = [("z0", z0.grad, zo.grad), ("path", path.grad, po.grad), ("t", t.grad, to.grad), ("W", func.linear.weight.grad, f64.linear.weight.grad)]
if with_times:
    pairs.append(("t_", t_.grad, t_o.grad))
for name, a, b in pairs:
    print(name, "max rel err", float((a.double().cpu() - b).abs().max() / b.abs().max()))
