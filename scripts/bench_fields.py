"""Timing of the non-linear vector fields on the full-size workload (B=32768, L=128, C=8, H=32), rk4:
Linear -> tanh (example/irregular_data.py) and Linear -> relu -> Linear -> tanh, width 128
(example/time_series_classification.py).  variant "generic" = VALU kernel (tanh field) / step-wise path (mlp).

    python scripts/bench_fields.py [--field tanh|mlp] [--adjoint] [--variants mfma,generic]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torchcde_amd as native  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--field", default="tanh", choices=["tanh", "linear", "mlp"])
    ap.add_argument("--adjoint", action="store_true")
    ap.add_argument("--variants", default="mfma,generic")
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--length", type=int, default=128)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=32)      # config 5's solve: --field mlp --length 65 --channels 14 --hidden 8
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, C, H = args.batch, args.length, args.channels, args.hidden
    x = make_series(B, L, C).to(dev)
    coeffs = native.hermite_cubic_coefficients_with_backward_differences(x)
    X = native.CubicSpline(coeffs)
    if args.field in ("tanh", "linear"):
        func = LinearField(H, C, scale=1.0 if args.field == "tanh" else 0.5, tanh=args.field == "tanh", seed=0).to(dev)
    else:
        class TwoLayer(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.linear1, self.linear2 = torch.nn.Linear(H, 128), torch.nn.Linear(128, H * C)

            def forward(self, t, z):
                return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], H, C)
        torch.manual_seed(0)
        func = TwoLayer().to(dev)
    z0 = torch.randn(B, H, device=dev)
    for variant in args.variants.split(","):
        def step():
            if args.adjoint:
                z = z0.clone().requires_grad_(True)
                out = native.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0), variant=variant)
                out[:, -1].sum().backward()
            else:
                with torch.no_grad():
                    native.cdeint(X, func, z0, X.interval, method="rk4", options=dict(step_size=1.0), variant=variant)
        if args.field == "mlp" and variant == "mfma":
            variant = "auto"
        reps = args.reps if variant != "generic" else 1
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"{args.field} field {'fwd+adjoint' if args.adjoint else 'forward'} variant={variant}: {ms:.2f} ms "
              f"({B / ms * 1e3 / 1e6:.2f} M series/s)", flush=True)


if __name__ == "__main__":
    main()
