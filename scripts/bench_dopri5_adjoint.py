#!/usr/bin/env python
"""BASELINE configs[3] shard with gradients: 32768 series, L = 128, LinearInterpolation, the reference's DEFAULT
call cdeint(X, func, z0, X.interval, options={'jump_t': knots}) (dopri5 + adjoint), loss.backward().  Reports the
fused forward (K4) and backward (K4a) times, their step counts, and -- optionally -- the step-wise path."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

front = sys.modules["torchcde_amd.cdeint"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, C, H = 128, 8, 32
dev = torch.device("cuda", 0)
x = make_series(B, L, C, seed=0).to(dev)
X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
func = LinearField(H, C, scale=0.25, seed=0).to(dev)
z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
res = {"B": B}
for variant in (["auto", "generic"] if "--stepwise" in sys.argv else ["auto"]):
    def fwd():
        z = z0.detach().requires_grad_(True)
        func.zero_grad()
        return cde.cdeint(X, func, z, X.interval, options=dict(jump_t=X.grid_points), variant=variant)
    out = fwd(); out[:, -1].sum().backward(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter()
    out[:, -1].sum().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tag = "fused" if variant == "auto" else "stepwise"
    res[tag + "_forward_ms"] = (t1 - t0) * 1e3
    res[tag + "_backward_ms"] = (t2 - t1) * 1e3
    if variant == "auto":
        res["forward_steps"] = dict(front.last_dopri5_stats)
        res["backward_steps"] = dict(front.last_dopri5_adjoint_stats)
print(json.dumps(res))
