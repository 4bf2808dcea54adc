"""The training call of the reference's example model (example/time_series_classification.py: two-layer field,
cdeint without `method`: dopri5 forward + adjoint) on the step-wise path.  python scripts/bench_default_call.py [B] [seminorm]"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as cde
from helpers import make_series
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
extra = dict(adjoint_options=dict(norm="seminorm")) if len(sys.argv) > 2 and sys.argv[2] == "seminorm" else {}
L, C, H = 128, 8, 32
dev = torch.device("cuda", 0)
class TwoLayer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.linear1, self.linear2 = torch.nn.Linear(H, 128), torch.nn.Linear(128, H * C)
        self.n = 0
    def forward(self, t, z):
        self.n += 1
        return self.linear2(self.linear1(z).relu()).tanh().view(*z.shape[:-1], H, C)
torch.manual_seed(0)
func = TwoLayer().to(dev)
x = make_series(B, L, C, seed=0).to(dev)
X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
z0 = torch.randn(B, H, device=dev)
for rep in range(2):
    func.n = 0
    z = z0.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = cde.cdeint(X, func, z, X.interval, **extra)  # the example's call: default dopri5, adjoint=True
    torch.cuda.synchronize(); t1 = time.perf_counter()
    nf = func.n
    out[:, -1].sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d forward %.3f s (%d evals), backward %.3f s (%d evals)" % (B, t1 - t0, nf, t2 - t1, func.n - nf), flush=True)
