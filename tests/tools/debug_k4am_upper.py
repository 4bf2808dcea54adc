"""Debug aid (round 6): the two-layer adaptive backward at 32 hidden units x 16 channels over several output times -- which
block of torchdiffeq's mixed adjoint norm decides the attempts whose error ratio differs from the float64 oracle's (the
oracle replays the kernel's attempts under a logging norm), per interval.
    python tests/tools/debug_k4am_upper.py [B] [C] [H]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native  # noqa: E402
from gpu_common import oracle_cde, oracle_interp, _TwoLayerField, make_series, DEV, _front  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H = int(sys.argv[3]) if len(sys.argv) > 3 else 24
L, width, kw = 6, 52, dict(rtol=1e-4, atol=1e-6)
front = _front()
x = make_series(B, L, C, seed=11)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(11))
t_out = torch.tensor([0., 2.2, 5.])
lw = torch.rand(B, 3, H, generator=torch.Generator().manual_seed(3)) + 0.5
func = _TwoLayerField(H, C, width, seed=3, final_tanh=True).to(DEV)
X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV)))
zd = z0.to(DEV).requires_grad_(True)
front.record_dopri5_steps = True
out = native.cdeint(X, func, zd, t_out.to(DEV), **kw)
print("dispatch:", front.last_dispatch()[0])
fwd = dict(front.last_dopri5_stats)
(out * lw.to(DEV)).sum().backward()
bwd = dict(front.last_dopri5_adjoint_stats)
front.record_dopri5_steps = False

f64 = _TwoLayerField(H, C, width, torch.float64, seed=3, final_tanh=True)
Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x.double()))
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
                        adjoint_options=dict(replay_attempts=[a.clone() for a in bwd["attempts"]], norm=norm), **kw)
(ref * lw.double()).sum().backward()
names = ["vjp_t", "y", "a", "W1", "b1", "W2", "b2"]
n_att = sum(len(a) for a in bwd["attempts"])
n_init = (len(log) - n_att) // len(bwd["attempts"])
print("norm calls", len(log), "attempts", n_att, "initial-step calls per interval", n_init)
pos = 0
for k, attempts in enumerate(bwd["attempts"]):
    init = log[pos:pos + n_init]
    pos += n_init
    rows = log[pos:pos + len(attempts)]
    pos += len(attempts)
    print("interval %d: initial-step norms (oracle):" % k)
    for vals in init:
        print("    ", ", ".join("%s %.4g" % (n, v) for v, n in sorted(zip(vals, names), reverse=True)[:3]))
    bad = 0
    for i, (row, vals) in enumerate(zip(attempts, rows)):
        mine, theirs = float(row[4]), max(vals)
        if abs(mine - theirs) > 0.02 * theirs + 0.01:
            bad += 1
            top = sorted(zip(vals, names), reverse=True)[:3]
            print("  attempt %4d  t0 %.4f dt %.5f acc %d  kernel %.4f oracle %.4f   oracle blocks: %s" % (
                i, row[0], row[1] - row[0], int(row[3]), mine, theirs, ", ".join("%s %.4f" % (n, v) for v, n in top)))
    print("  %d of %d attempts outside the band" % (bad, len(attempts)))
for (name, got), want in zip(func.named_parameters(), f64.parameters()):
    print(name, "max rel err", float((got.grad.double().cpu() - want.grad).abs().max() / want.grad.abs().max()))
print("dL/dz0 max rel err", float((zd.grad.double().cpu() - zo.grad).abs().max() / zo.grad.abs().max()))
