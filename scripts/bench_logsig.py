"""BASELINE.json configs[4] pre-processing on one GPU: logsig_windows(depth 3) of 32768 series, L=512, 3 channels
(-> 14 logsignature channels), window length 8 -> 65 points per series; then the linear-interpolation coefficients."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchcde_amd as cde

B, L, C, depth, window = 32768, 512, 3, 3, 8.0
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
x = (torch.randn(B, L, C, generator=gen) * 0.1).cumsum(1)
x[..., 0] = torch.linspace(0, 1, L)
x = x.to(dev)
out = cde.logsig_windows(x, depth, window); torch.cuda.synchronize()
times = []
for _ in range(5):
    t0 = time.perf_counter(); out = cde.logsig_windows(x, depth, window); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
best = min(times)
print(json.dumps({"config": "logsig_windows depth=%d, B=%d L=%d C=%d window=%g" % (depth, B, L, C, window),
                  "out_shape": list(out.shape), "seconds": best, "series_per_s": B / best,
                  "input_GBs": x.numel() * 4 / best / 1e9}))
# the same transform with gradients w.r.t. the raw data (cde_logsig_windows_backward)
xg = x.clone().requires_grad_(True)
w = torch.randn_like(out)
def fwd_bwd():
    xg.grad = None
    (cde.logsig_windows(xg, depth, window) * w).sum().backward()
fwd_bwd(); torch.cuda.synchronize()
times = []
for _ in range(5):
    t0 = time.perf_counter(); fwd_bwd(); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
print(json.dumps({"config": "the same, forward + backward w.r.t. x", "seconds": min(times), "series_per_s": B / min(times)}))
