import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torchcde_amd as cde
from helpers import TwoLayerField, make_series
dev = "cuda"
for B in (64, 4096):
    x = make_series(B, 128, 8, seed=0).to(dev)
    X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
    f = TwoLayerField(32, 8, 128, seed=0).to(dev)
    z0 = torch.randn(B, 32, device=dev)
    with torch.no_grad():
        for _ in range(3): cde.cdeint(X, f, z0, X.interval, method="rk4", options=dict(step_size=1.0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): cde.cdeint(X, f, z0, X.interval, method="rk4", options=dict(step_size=1.0))
        torch.cuda.synchronize(); print("B", B, "rk4 two-layer forward ms", (time.perf_counter() - t0) * 100)
