"""Small driver for rocprofv3 counter passes: the bench workload (B=32768, L=128, C=8, H=32), a few
forward+adjoint steps and a few Hermite fits -- nothing else.  Usage (one PMC set per run):
    rocprofv3 --pmc <counters> --kernel-trace -d DIR -o NAME -- python scripts/prof_workload.py [steps] [batch] [variant] [adjoint|backprop]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import torchcde_amd as cde  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.B
variant = sys.argv[3] if len(sys.argv) > 3 else "auto"
adjoint = not (len(sys.argv) > 4 and sys.argv[4] == "backprop")      # "backprop": adjoint=False (K2 with stage stores + K3d)
dev = torch.device("cuda", 0)
x, func, z0 = bench.make_workload(dev, seed=0)
x, z0 = x[:batch].contiguous(), z0[:batch].contiguous()
for _ in range(steps):
    coeffs = cde.hermite_cubic_coefficients_with_backward_differences(x)
X = cde.CubicSpline(coeffs)
for _ in range(steps):
    z = z0.detach().requires_grad_(True)
    func.zero_grad()
    out = cde.cdeint(X, func, z, X.interval, method="rk4", options={"step_size": 1.0}, variant=variant, adjoint=adjoint)
    out[:, -1].sum().backward()
torch.cuda.synchronize()
print("done")
