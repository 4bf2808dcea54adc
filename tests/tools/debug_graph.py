import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as native
from torchcde_amd import fields
from helpers import LinearField, make_series
cd = importlib.import_module("torchcde_amd.cdeint")
DEV = "cuda"
for (H, C, act) in ((64, 8, False), (32, 16, True)):
    B, L = 70, 9
    x = make_series(B, L, C, torch.float32, seed=300 + H).to(DEV)
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x))
    func = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=5).to(DEV)
    gen = torch.Generator().manual_seed(H * C)
    z0 = torch.randn(B, H, generator=gen).to(DEV)
    t_out = torch.tensor([0., 3.5, 8.])
    go = (torch.rand(B, 3, H, generator=gen) + 0.5).to(DEV)
    field, _ = fields.probe(func, t_out[0].to(DEV), z0)
    plan = cd._Plan(X, field, (B,), H, C, t_out, 1.0, 1.0, True, cd._lib.VARIANT_AUTO)
    w, b = func.linear.weight.detach(), func.linear.bias.detach()
    def step():
        out = plan.run_forward(z0, w, b)
        gz, gw, gb, _ = plan.run_adjoint(out, go, w, b)
        return out, gz, gw.clone(), gb.clone()
    ref = step(); torch.cuda.synchronize()
    names = ["out", "gz", "gw", "gb"]
    for rep in range(6):
        # poison free memory so that uninitialised reads show
        junk = [torch.full((1 << 22,), float(1e30) * (rep + 1), device=DEV) for _ in range(8)]
        del junk
        cur = step(); torch.cuda.synchronize()
        print(H, C, "eager rep", rep, [bool(torch.equal(a, b_)) for a, b_ in zip(cur, ref)],
              [float((a - b_).abs().max()) for a, b_ in zip(cur, ref)])

print("---- graph capture")
for (H, C, act) in ((64, 8, False), (32, 16, True), (32, 8, False)):
    B, L = 70, 9
    x = make_series(B, L, C, torch.float32, seed=300 + H).to(DEV)
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x))
    func = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=5).to(DEV)
    gen = torch.Generator().manual_seed(H * C)
    z0 = torch.randn(B, H, generator=gen).to(DEV)
    t_out = torch.tensor([0., 3.5, 8.])
    go = (torch.rand(B, 3, H, generator=gen) + 0.5).to(DEV)
    field, _ = fields.probe(func, t_out[0].to(DEV), z0)
    plan = cd._Plan(X, field, (B,), H, C, t_out, 1.0, 1.0, True, cd._lib.VARIANT_AUTO)
    w, b = func.linear.weight.detach(), func.linear.bias.detach()
    def step():
        out = plan.run_forward(z0, w, b)
        gz, gw, gb, _ = plan.run_adjoint(out, go, w, b)
        return out, gz, gw.clone(), gb.clone()
    ref = step(); torch.cuda.synchronize()
    for mode in ("nofill", "fill"):
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(); side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                captured = step()
        torch.cuda.current_stream().wait_stream(side)
        for rep in range(3):
            if mode == "fill":
                for tns in captured:
                    tns.fill_(float("nan"))
            graph.replay(); torch.cuda.synchronize()
            print(H, C, mode, "replay", rep, [bool(torch.equal(a, b_)) for a, b_ in zip(captured, ref)],
                  ["%.3g" % float((a - b_).abs().max()) for a, b_ in zip(captured, ref)])
