#!/usr/bin/env python
"""Per-GPU batch sweep of the headline step (cdeint RK4 forward + adjoint backward) on ONE GPU: the 1-GPU proxy of
the strong-scaling experiment (B_total = 32768 over N GPUs leaves 32768/N series per GPU).

    python scripts/bench_strong.py [--batches 4096,8192,16384,32768] [--steps 20] [--variant auto|mfma|split]

Prints one JSON line per batch: wall ms per step (host loop, synchronised at both ends), HIP-event ms of the forward
and adjoint C-ABI calls, and the host-side enqueue time of a step (how long Python needs to queue it)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

L, C, H = 128, 8, 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4096,8192,16384,32768")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variant", default="auto")
    args = ap.parse_args()
    import torchcde_amd as cde
    front_events = sys.modules["torchcde_amd.cdeint"]      # its `event_log` attribute is this thread's
    from helpers import LinearField, make_series
    cde.load()
    dev = torch.device("cuda", 0)
    func = LinearField(H, C, scale=0.25, seed=0).to(dev)
    params = list(func.parameters())
    for B in [int(b) for b in args.batches.split(",")]:
        x = make_series(B, L, C, seed=0).to(dev)
        z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(dev)
        X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
        t = X.interval

        def step():
            z = z0.detach().requires_grad_(True)
            for p in params:
                p.grad = None
            out = cde.cdeint(X, func, z, t, method="rk4", options={"step_size": 1.0}, variant=args.variant)
            out[:, -1].sum().backward()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        front_events.event_log = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_enqueued = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        log, front_events.event_log = front_events.event_log, None
        fwd = [a.elapsed_time(b) for k, a, b in log if k == "forward"]
        adj = [a.elapsed_time(b) for k, a, b in log if k == "adjoint"]
        print(json.dumps({"B": B, "variant": args.variant, "wall_ms_per_step": wall / args.steps * 1e3,
                          "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
                          "forward_ms": sum(fwd) / len(fwd), "adjoint_ms": sum(adj) / len(adj),
                          "series_per_s": B * args.steps / wall}), flush=True)


if __name__ == "__main__":
    main()
