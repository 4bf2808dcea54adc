"""adjoint=False on the headline workload with a coefficient tensor that requires a gradient (cde_rk4_backprop_linear_dcontrol):
ms per forward + backward step and the two kernels' own durations (HIP events on the launching stream)."""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as cde
from helpers import LinearField, make_series
front = sys.modules["torchcde_amd.cdeint"]
B, L, C, H = 32768, 128, 8, 32
x = make_series(B, L, C, seed=0).cuda(); z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).cuda()
base = cde.hermite_cubic_coefficients_with_backward_differences(x)
for tanh in (False, True):
    func = LinearField(H, C, scale=0.25, tanh=tanh, seed=0).cuda()
    for control in (False, True):
        coeffs = base.clone().requires_grad_(control)
        X = cde.CubicSpline(coeffs)
        def step():
            z = z0.detach().requires_grad_(True); func.zero_grad(); coeffs.grad = None
            cde.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0), adjoint=False)[:, -1].sum().backward()
        for _ in range(4): step()
        torch.cuda.synchronize(); front.event_log = []
        t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
        log, front.event_log = front.event_log, None
        print("tanh=%s control gradients=%s: step %.3f ms  forward %.3f  backward %.3f" % (tanh, control, wall,
              sum(a.elapsed_time(b) for k, a, b in log if k == "forward") / 10, sum(a.elapsed_time(b) for k, a, b in log if k == "backprop") / 10))
