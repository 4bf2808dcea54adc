"""BASELINE.json configs[3] on ONE GPU's shard: 32768 series, L=128, C=8, H=32, linear_interpolation_coeffs +
LinearInterpolation control, adaptive dopri5 (rtol 1e-4, atol 1e-6, jump_t = knots as README.md:194-200 prescribes),
forward only.  Prints one JSON line; `--variant generic|auto` selects the attempt kernel."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde
from helpers import LinearField, make_series

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="auto")
ap.add_argument("--batch", type=int, default=32768)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--field", default="linear", choices=["linear", "tanh", "two_layer"])
ap.add_argument("--hidden", type=int, default=32)
ap.add_argument("--channels", type=int, default=8)
a = ap.parse_args()
B, L, C, H = a.batch, 128, a.channels, a.hidden
dev = torch.device("cuda", 0)
x = make_series(B, L, C, seed=0).to(dev)
if a.field == "two_layer":
    class TwoLayer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear1, self.linear2 = torch.nn.Linear(H, 128), torch.nn.Linear(128, H * C)

        def forward(self, t, z):
            return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], H, C)
    torch.manual_seed(0)
    func = TwoLayer().to(dev)
else:
    func = LinearField(H, C, scale=0.25 if a.field == "linear" else 1.0, tanh=a.field == "tanh", seed=0).to(dev)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
import importlib
mod = importlib.import_module("torchcde_amd.cdeint")

def solve():
    with torch.no_grad():
        return cde.cdeint(X, func, z0, X.interval, method="dopri5", rtol=1e-4, atol=1e-6,
                          options=dict(jump_t=X.grid_points), variant=a.variant)
out = solve(); torch.cuda.synchronize()
times = []
for _ in range(a.repeats):
    t0 = time.perf_counter(); out = solve(); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
st = dict(mod.last_dopri5_stats)
best = min(times)
evals = 6 * (st["n_accept"] + st["n_reject"]) + st["n_accept"] + 2     # stages + post-jump refreshes + initial step
print(json.dumps({"field": a.field, "config": "dopri5 + LinearInterpolation, B=%d L=%d C=%d H=%d, rtol 1e-4 atol 1e-6, jump_t=knots" % (B, L, C, H),
                  "variant": a.variant, "seconds": best, "series_per_s": B / best, "stats": st,
                  "us_per_attempt_launch": best / st["launches"] * 1e6,
                  "field_evals": evals, "tflops": evals * B * (74240 if a.field == "two_layer" else 2 * H * C * H + 2 * H * C) / best / 1e12,
                  "finite": bool(torch.isfinite(out).all())}))
