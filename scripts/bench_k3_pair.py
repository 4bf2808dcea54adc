"""K3j (one wave per tile) against K3p (chain + helper wave per tile, rk4_adjoint_pair.hip) on the headline workload:
bitwise comparison of every gradient and the adjoint kernel's duration (HIP events around the C-ABI call).
    python scripts/bench_k3_pair.py [series=32768] [reps=10]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
front = sys.modules["torchcde_amd.cdeint"]      # (the attribute `torchcde_amd.cdeint` is the function)
from helpers import LinearField, make_series  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L, C, H = 128, 8, 32
dev = "cuda"
x = make_series(B, L, C, seed=0).to(dev)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
res = {}
for waves in ("1", "2"):
    cde.set_option("k3_waves", int(waves))
    func = LinearField(H, C, scale=0.25, seed=0).to(dev)

    def step():
        z = z0.detach().requires_grad_(True)
        func.zero_grad()
        out = cde.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0), variant="mfma")
        out[:, -1].sum().backward()
        return z.grad, func.linear.weight.grad.clone(), func.linear.bias.grad.clone()
    for _ in range(3):
        g = step()
    torch.cuda.synchronize()
    front.event_log = []
    t0 = time.perf_counter()
    for _ in range(reps):
        g = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    log, front.event_log = front.event_log, None
    adj = [a.elapsed_time(b) for kind, a, b in log if kind == "adjoint"]
    fwd = [a.elapsed_time(b) for kind, a, b in log if kind == "forward"]
    res[waves] = g
    print("CDE_K3_WAVES=%s  B=%d  step %.3f ms  forward %.3f ms  adjoint %.3f ms (min %.3f)"
          % (waves, B, wall, sum(fwd) / len(fwd), sum(adj) / len(adj), min(adj)), flush=True)
same = all(torch.equal(a, b) for a, b in zip(res["1"], res["2"]))
print("bitwise equal gradients:", same)
if not same:
    for name, a, b in zip(("dz0", "dW", "db"), res["1"], res["2"]):
        print(name, "max abs diff %.3g of %.3g" % ((a - b).abs().max().item(), a.abs().max().item()))
