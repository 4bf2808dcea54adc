"""rocprofv3 driver: config-4 shard, default dopri5 + adjoint call, one repetition.  argv: [norm=mixed|seminorm] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

norm = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda", 0)
x = make_series(B, 128, 8, seed=0).to(dev)
X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
func = LinearField(32, 8, scale=0.25, seed=0).to(dev)
z = torch.randn(B, 32, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
extra = dict(adjoint_options=dict(norm="seminorm", jump_t=X.grid_points)) if norm == "seminorm" else {}
out = cde.cdeint(X, func, z, X.interval, options=dict(jump_t=X.grid_points), **extra)
out[:, -1].sum().backward()
torch.cuda.synchronize()
print("done")
