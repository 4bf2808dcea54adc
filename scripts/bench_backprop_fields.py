"""adjoint=False (reverse mode through the rk4 steps) for the three fused field families at the benchmark size:
    python scripts/bench_backprop_fields.py [series=32768]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, TwoLayerField, make_series  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, C, H = 128, 8, 32
dev = "cuda"
x = make_series(B, L, C, seed=0).to(dev)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
fields = {"affine": LinearField(H, C, scale=0.25, seed=0), "tanh": LinearField(H, C, scale=0.5, tanh=True, seed=2),
          "two_layer": TwoLayerField(H, C, 128, seed=2)}
for name, func in fields.items():
    func = func.to(dev)
    for adjoint in (True, False):
        def step():
            z = z0.detach().requires_grad_(True)
            func.zero_grad()
            cde.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0), adjoint=adjoint)[:, -1].sum().backward()
        step(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%-10s adjoint=%-5s forward + backward %.2f ms" % (name, adjoint, best * 1e3), flush=True)
