"""Debug aid: per-launch controller trace of the native dopri5 vs the oracle's attempt sequence."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde
from torchcde_amd import _lib
import importlib
C = importlib.import_module("torchcde_amd.cdeint")
from oracle import odeint as oo, interp as oi, cde as oc
from helpers import LinearField, make_series

dtype = torch.float64
B, L, Cc, H = 10, 9, 3, 5
x = make_series(B, L, Cc, dtype, seed=5)
coeffs = oi.hermite_bdiff_coeffs(x)
func = LinearField(H, Cc, dtype, scale=0.5, tanh=True, seed=8)
z0 = torch.randn(B, H, dtype=dtype, generator=torch.Generator().manual_seed(8))
Xo = oi.CubicPath(coeffs)
field = oo._Field(oc.ControlledField(Xo, func))
solver = oo._Dopri5(field, z0, 1e-8, 1e-10, oo._rms)
# instrument the oracle
trace = []
orig = solver._next_dt
def spy(last, ratio):
    trace.append((float(last), float(ratio)))
    return orig(last, ratio)
solver._next_dt = spy
with torch.no_grad():
    ref = solver.integrate(Xo.interval)
print("oracle attempts", len(trace), "accept", solver.n_accept, "reject", solver.n_reject)

C._DOPRI_CHUNK = 1
dfunc = LinearField(H, Cc, dtype, scale=0.5, tanh=True, seed=8).cuda()
X = cde.CubicSpline(coeffs.cuda())
lib = _lib.load()
from torchcde_amd.fields import probe
fld, _ = probe(dfunc, X.interval[0], z0.cuda())
plan = C._Dopri5Plan(X, fld, (B,), H, Cc, X.interval, 1e-8, 1e-10, None)
# manual loop with chunk 1 to print the controller after every launch
out = torch.empty(B, 2, H, dtype=dtype, device="cuda")
ws = torch.empty(lib.cde_dopri5_workspace_bytes(B, Cc, H, 1), dtype=torch.uint8, device="cuda")
w, b = fld.weight.detach().contiguous(), fld.bias.detach().contiguous()
z0c = z0.cuda().contiguous()
size = ctypes.sizeof(_lib.DopriStatus)
native = []
for launched in range(0, 400):
    _lib.check(lib.cde_dopri5_advance(_lib.ptr(plan.coeffs), _lib.ptr(plan.knots), plan.n_intervals, plan.degree, _lib.ptr(w), _lib.ptr(b),
        plan.act, _lib.ptr(z0c), _lib.ptr(plan.t_out), plan.n_out, _lib.ptr(None), 0, 1e-8, 1e-10, 0.9, 10.0, 0.2, _lib.ptr(out), B, Cc, H, 1, 0,
        _lib.ptr(ws), ws.numel(), launched, 1, _lib.stream_ptr(out.device)), "adv")
    k = (launched + 1) & 1
    st = _lib.DopriStatus.from_buffer_copy(ws[k * size:(k + 1) * size].cpu().numpy().tobytes())
    native.append((st.phase, st.t_hi, st.dt, st.dt_try, st.n_accept, st.n_reject, st.h0))
    if st.phase == 4:
        break
print("native launches", len(native))
for i, r in enumerate(native[:8]): print(i, r)
print("oracle first attempts (dt, ratio):")
for r in trace[:6]: print(r)
# native attempts: dt_try recorded at launches >= 2
nat_dt = [r[3] for r in native[2:]]
for i, (a, b_) in enumerate(zip(nat_dt, [t[0] for t in trace])):
    if abs(a - b_) > 1e-12 * max(1, abs(b_)):
        print("first dt mismatch at attempt", i, a, b_); break
else:
    print("all attempted dt equal")
print("max |out-ref|", (out.cpu() - ref.permute(1, 0, 2)).abs().max().item())
