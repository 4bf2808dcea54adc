"""rocprofv3 driver for the non-linear vector fields (same workload shape as the bench: B=32768, L=128, C=8, H=32):
Linear -> tanh and Linear -> relu -> Linear -> tanh, a few forward + adjoint solves each.
    rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python scripts/prof_fields.py [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
x, _, z0 = bench.make_workload(dev, seed=0)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
H, C = bench.H, bench.C


class TwoLayer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.linear1, self.linear2 = torch.nn.Linear(H, 128), torch.nn.Linear(128, H * C)

    def forward(self, t, z):
        return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], H, C)


torch.manual_seed(0)
for func in (LinearField(H, C, scale=1.0, tanh=True, seed=0).to(dev), TwoLayer().to(dev)):
    for _ in range(reps):
        z = z0.detach().requires_grad_(True)
        cde.cdeint(X, func, z, X.interval, method="rk4", options={"step_size": 1.0})[:, -1].sum().backward()
torch.cuda.synchronize()
print("done")
