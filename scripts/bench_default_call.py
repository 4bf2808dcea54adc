"""The training call of the reference's example model (example/time_series_classification.py: two-layer field,
cdeint without `method`: dopri5 forward + adjoint).  Fused since round 3 (K4 forward, K4am backward); `stepwise` as a
third argument forces the host-driven path of round 2.
    python scripts/bench_default_call.py [B] [seminorm|mixed] [stepwise|fused] [C] [H]     (C, H: e.g. 14 8 = config 5's shape)"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torchcde_amd as cde
from helpers import make_series
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
extra = dict(adjoint_options=dict(norm="seminorm")) if len(sys.argv) > 2 and sys.argv[2] == "seminorm" else {}
if len(sys.argv) > 3 and sys.argv[3] == "stepwise":
    extra["variant"] = "generic"
L = 128
C = int(sys.argv[4]) if len(sys.argv) > 4 else 8
H = int(sys.argv[5]) if len(sys.argv) > 5 else 32
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
    path = type(out.grad_fn).__name__
    torch.cuda.synchronize(); t1 = time.perf_counter()
    nf = func.n
    out[:, -1].sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    front = sys.modules["torchcde_amd.cdeint"]
    st = dict(front.last_dopri5_adjoint_stats) if path == "_FusedMlpDopri5Backward" else {}
    attempts = st.get("n_accept", 0) + st.get("n_reject", 0)
    print("B=%d %s forward %.4f s (%s), backward %.4f s (%d accepted + %d rejected attempts, %.1f us per attempt)"
          % (B, path, t1 - t0, dict(front.last_dopri5_stats) if st else "%d evals" % nf, t2 - t1, st.get("n_accept", 0),
             st.get("n_reject", 0), (t2 - t1) * 1e6 / max(attempts, 1)), flush=True)
