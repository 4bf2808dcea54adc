"""rocprofv3 driver: the reference examples' training call (two-layer field, default dopri5 + adjoint) -- K4 forward,
K4am backward.    rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python scripts/prof_default_mlp.py [batch] [norm]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import TwoLayerField, make_series  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
extra = dict(adjoint_options=dict(norm="seminorm")) if len(sys.argv) > 2 and sys.argv[2] == "seminorm" else {}
dev = torch.device("cuda", 0)
x = make_series(B, 128, 8, seed=0).to(dev)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
func = TwoLayerField(32, 8, 128, seed=0).to(dev)
z = torch.randn(B, 32, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
out = cde.cdeint(X, func, z, X.interval, **extra)
assert type(out.grad_fn).__name__ == "_FusedMlpDopri5Backward"
out[:, -1].sum().backward()
torch.cuda.synchronize()
print("done")
