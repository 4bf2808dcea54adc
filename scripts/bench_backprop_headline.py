import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as cde
from helpers import LinearField, make_series
front = sys.modules["torchcde_amd.cdeint"]
B, L, C, H = 32768, 128, 8, 32
x = make_series(B, L, C, seed=0).cuda(); z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).cuda()
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
func = LinearField(H, C, scale=0.25, seed=0).cuda()
def step():
    z = z0.detach().requires_grad_(True); func.zero_grad()
    cde.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0), adjoint=False)[:, -1].sum().backward()
for _ in range(3): step()
torch.cuda.synchronize(); front.event_log = []
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
log, front.event_log = front.event_log, None
print("adjoint=False: step %.3f ms  forward %.3f  backward %.3f" % (wall, sum(a.elapsed_time(b) for k,a,b in log if k=="forward")/10, sum(a.elapsed_time(b) for k,a,b in log if k=="backprop")/10))
