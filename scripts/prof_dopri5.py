"""rocprofv3 driver: BASELINE configs[3] shard with gradients (default dopri5 + adjoint call), a few repetitions.
    rocprofv3 --pmc <counters> --kernel-trace -d DIR -o NAME -- python scripts/prof_dopri5.py [reps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda", 0)
x = make_series(B, 128, 8, seed=0).to(dev)
X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
func = LinearField(32, 8, scale=0.25, seed=0).to(dev)
z0 = torch.randn(B, 32, generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(reps):
    z = z0.detach().requires_grad_(True)
    func.zero_grad()
    out = cde.cdeint(X, func, z, X.interval, options=dict(jump_t=X.grid_points))
    out[:, -1].sum().backward()
torch.cuda.synchronize()
print("done")
