import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torchcde_amd as cde
from helpers import LinearField, make_series
dev = "cuda"
for B in (64, 4096):
    x = make_series(B, 128, 8, seed=0).to(dev)
    X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
    f = LinearField(32, 8, scale=0.5, tanh=True, seed=0).to(dev)
    z0 = torch.randn(B, 32, device=dev)
    with torch.no_grad():
        for _ in range(2): out = cde.cdeint(X, f, z0, X.interval)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = cde.cdeint(X, f, z0, X.interval)
        torch.cuda.synchronize(); print("B", B, "tanh dopri5 forward ms", round((time.perf_counter() - t0) * 200, 3), float(out.abs().sum()))
