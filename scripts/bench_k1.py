#!/usr/bin/env python
"""K1 (Hermite fit) timing at the benchmark size: per-call time of the public function (default unit-spaced t and an
irregular t), HBM rate against the algorithmic 20,352 B per series.  python scripts/bench_k1.py [B]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from helpers import make_series  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, C = 128, 8
dev = torch.device("cuda", 0)
x = make_series(B, L, C, seed=0).to(dev)
t_irr = (torch.rand(L, dtype=torch.float64).cumsum(0) + 0.1).float().to(dev)
res = {"B": B}
for name, t in (("unit_t", None), ("irregular_t", t_irr)):
    for _ in range(3):
        cde.hermite_cubic_coefficients_with_backward_differences(x, t)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        cde.hermite_cubic_coefficients_with_backward_differences(x, t)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    res[name + "_ms_per_call"] = ms
    res[name + "_hbm_gbs"] = B * 20352 / (ms * 1e-3) / 1e9
print(json.dumps(res))
