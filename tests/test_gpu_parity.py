"""GPU parity tests: the HIP path (through the ctypes C ABI) against the CPU oracle, the committed
golden fixtures generated from the reference, and size-independent properties at full benchmark size.

Tolerances (stated once, used below):
  * interval indices: bit-exact (int64 equality)
  * coefficients / frac / spline value & slope: bit-exact against the reference's floats
  * trajectories: rtol 1e-4, atol 1e-6 (the north star's bar) in float32; 1e-9 / 1e-11 in float64.
    At the full 127-step length float32 round-off accumulated over 508 stages reaches a few 1e-6 absolute on
    O(1) states (the CPU float32 oracle deviates from float64 by the same amount, which the test measures), so
    there atol is 1e-5 and the kernel's error is additionally bounded by 4x the CPU-float32 error.
  * gradients: rtol 1e-3 (float32 kernels vs float64 oracle), 1e-8 in float64
"""

import pytest
import torch

from oracle import cde as oracle_cde, interp as oracle_interp
from helpers import LinearField, golden_field, make_series

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, rtol, atol):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        err = ((a - b).abs() / (atol + rtol * b.abs())).max().item()
        raise AssertionError("mismatch: worst error = %.3g x tolerance, max abs diff %.3g" % (err, (a - b).abs().max()))


# =========================================================================================== K1 / K1b
def test_hermite_coefficients_bit_exact_vs_reference_golden(native, golden_interp):
    for case in golden_interp:
        x = case["x"].to(DEV)
        t = None if case["t"] is None else case["t"].to(DEV)
        got = native.hermite_cubic_coefficients_with_backward_differences(x, t)
        assert got.shape == case["coeffs"].shape and got.dtype == case["coeffs"].dtype
        assert torch.equal(got.cpu(), case["coeffs"]), "coeffs differ from the reference's floats"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("C", [1, 3, 8])
def test_hermite_coefficients_vs_oracle_seeded(native, dtype, C):
    gen = torch.Generator().manual_seed(100 + C)
    for batch, L in (((7,), 2), ((2, 3), 33), ((257,), 128)):
        x = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
        t = (torch.rand(L, generator=gen, dtype=torch.float64) + 0.05).cumsum(0).to(dtype)
        for tt in (None, t):
            got = native.hermite_cubic_coefficients_with_backward_differences(
                x.to(DEV), None if tt is None else tt.to(DEV))
            assert torch.equal(got.cpu(), oracle_interp.hermite_bdiff_coeffs(x, tt))


def test_linear_coeffs_returns_same_tensor(native):
    x = torch.randn(3, 6, 2, device=DEV)
    assert native.linear_interpolation_coeffs(x) is x
    with pytest.raises(NotImplementedError, match="rectilinear"):
        native.linear_interpolation_coeffs(x, rectilinear=0)


def test_missing_values_bit_exact_vs_reference_golden(native):
    """NaN-aware construction (interpolation_linear.py:13-84) and the Hermite fit on top of it: bit-exact against
    fixtures from the reference, incl. leading/trailing gaps and all-NaN channels."""
    import os
    from conftest import GOLDEN
    cases = torch.load(os.path.join(GOLDEN, "nan_fill.pt"))
    for case in cases:
        x = case["x"].to(DEV)
        t = None if case["t"] is None else case["t"].to(DEV)
        filled = native.linear_interpolation_coeffs(x, t)
        assert filled is not x and not torch.isnan(filled).any()
        assert torch.equal(filled.cpu(), case["filled"])
        assert torch.equal(native.hermite_cubic_coefficients_with_backward_differences(x, t).cpu(), case["hermite"])


def test_missing_values_large_batch_vs_oracle(native):
    B, L, C = 2048, 64, 8
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, L, C, generator=gen)
    x = x.masked_fill(torch.rand(B, L, C, generator=gen) < 0.1, float("nan"))      # the survey's 10 % NaN workload
    got = native.linear_interpolation_coeffs(x.to(DEV)).cpu()
    sample = torch.arange(0, B, 97)
    assert torch.equal(got[sample], oracle_interp.linear_coeffs(x[sample]))
    keep = ~torch.isnan(x)
    assert torch.equal(got[keep], x[keep]) and not torch.isnan(got).any()          # observations untouched everywhere


def test_interpret_t_evaluate_derivative_bit_exact_vs_reference_golden(native, golden_interp):
    for case in golden_interp:
        knots = case["knots"].to(DEV)
        explicit = case["t"] is not None
        X = native.CubicSpline(case["coeffs"].to(DEV), knots if explicit else None)
        assert torch.equal(X.grid_points.cpu(), case["knots"])
        tq = case["tq"].to(DEV)
        frac, index = X._interpret_t(tq)
        assert index.dtype == torch.int64 and torch.equal(index.cpu(), case["index"])      # bit-exact indices
        assert torch.equal(frac.cpu(), case["frac"])
        value, slope = X.evaluate(tq), X.derivative(tq)
        assert value.shape == case["value"].shape
        assert torch.equal(value.cpu(), case["value"]) and torch.equal(slope.cpu(), case["slope"])
        # scalar-time call (what cdeint's callers use: X.evaluate(X.interval[0]))
        assert torch.equal(X.evaluate(tq[3]).cpu(), case["value"][..., 3, :])
        L = native.LinearInterpolation(case["x"].to(DEV), knots if explicit else None)
        lfrac, lindex = L._interpret_t(tq)
        assert torch.equal(lindex.cpu(), case["lin_index"]) and torch.equal(lfrac.cpu(), case["lin_frac"])
        assert torch.equal(L.evaluate(tq).cpu(), case["lin_value"])
        assert torch.equal(L.derivative(tq).cpu(), case["lin_slope"])


def test_path_output_shapes_follow_reference_contract(native):
    """batch_dims + t.shape + (channels,)  (reference test_natural_cubic_spline.py:151-167)"""
    coeffs = torch.randn(2, 3, 7, 12, device=DEV)
    X = native.CubicSpline(coeffs)
    for tshape in ((), (5,), (2, 4)):
        t = torch.rand(tshape, device=DEV) * 7
        assert X.evaluate(t).shape == (2, 3) + tshape + (3,)
        assert X.derivative(t).shape == (2, 3) + tshape + (3,)


# =========================================================================================== cdeint
def _run_native(native, case, variant, adjoint):
    func = golden_field(case).to(DEV)
    X = native.CubicSpline(case["coeffs"].to(DEV), None if case["knots"] is None else case["knots"].to(DEV))
    z0 = case["z0"].to(DEV).requires_grad_(True)
    out = native.cdeint(X, func, z0, case["t_out"].to(DEV), adjoint=adjoint, method=case["method"],
                        options=case["options"], variant=variant)
    return func, z0, out


@pytest.mark.parametrize("variant", ["generic", "auto"])
def test_cdeint_rk4_vs_reference_golden(native, golden_cde, variant):
    """Trajectories and adjoint gradients against the fixtures produced by the reference's solver.py
    (driven by the oracle integrator).  README toy (config 1) included."""
    ran = 0
    for case in golden_cde:
        if case["method"] != "rk4":
            continue
        f64 = case["z0"].dtype == torch.float64
        rt, at = (1e-9, 1e-11) if f64 else (1e-4, 1e-6)
        func, z0, out = _run_native(native, case, variant, adjoint=True)
        assert out.shape == case["out_adjoint"].shape
        _close(out, case["out_adjoint"], rt, at)
        w = torch.linspace(0.5, 1.5, out.numel(), dtype=out.dtype, device=DEV).view_as(out)
        (out * w).sum().backward()
        grt, gat = (1e-8, 1e-10) if f64 else (1e-3, 1e-4)     # atol relative to the largest gradient entry
        for got, ref in ((z0.grad, case["gz0_adjoint"]), (func.linear.weight.grad, case["gW_adjoint"]),
                         (func.linear.bias.grad, case["gb_adjoint"])):
            _close(got, ref, grt, gat * ref.abs().max().item())
        ran += 1
    assert ran >= 5


def test_stage_table_is_bit_exact_with_reference_interpret_t(native):
    """Every stage time of every RK4 step must resolve to the interval index (int64, bit-exact) and
    fractional part that CubicSpline._interpret_t gives for torchdiffeq's stage times."""
    from torchcde_amd.cdeint import _Plan, _fixed_grid
    from torchcde_amd.fields import probe
    for dtype, L, step, explicit in ((torch.float32, 128, 1.0, False), (torch.float32, 40, 0.3, True),
                                     (torch.float64, 17, 0.7, True)):
        x = make_series(4, L, 8, dtype, seed=3)
        knots = None
        if explicit:
            knots = (torch.rand(L, dtype=torch.float64).cumsum(0) + 0.2).to(dtype)
        coeffs = oracle_interp.hermite_bdiff_coeffs(x, knots)
        X = native.CubicSpline(coeffs.to(DEV), None if knots is None else knots.to(DEV))
        func = LinearField(32, 8, dtype, scale=0.25).to(DEV)
        z0 = torch.randn(4, 32, dtype=dtype, device=DEV)
        field, _ = probe(func, X.interval[0], z0)
        plan = _Plan(X, field, (4,), 32, 8, X.interval, step, step, True, 1)
        plan.run_forward(z0, field.weight, field.bias)
        grid = _fixed_grid(X.interval.cpu(), step)
        cpu_knots = X.grid_points.cpu()
        exp_idx, exp_frac = [], []
        third, two_thirds = 1 / 3, 2 / 3
        for t0, t1 in zip(grid[:-1], grid[1:]):
            dt = t1 - t0
            for ts in (t0, t0 + dt * third, t0 + dt * two_thirds, t1):
                frac, idx = oracle_interp.locate(ts.to(dtype), cpu_knots, coeffs.size(-2), dtype, "cpu")
                exp_idx.append(idx)
                exp_frac.append(frac)
        assert torch.equal(plan.stage_index.cpu(), torch.stack(exp_idx))
        assert torch.equal(plan.stage_frac.cpu(), torch.stack(exp_frac))


def _oracle_solution(coeffs, knots, func, z0, t_out, step, loss_weight=None):
    """float64 oracle: trajectories and adjoint gradients."""
    f64 = LinearField(func.H, func.C, torch.float64, tanh=func.tanh)
    with torch.no_grad():
        f64.linear.weight.copy_(func.linear.weight.double().cpu())
        f64.linear.bias.copy_(func.linear.bias.double().cpu())
    X = oracle_interp.CubicPath(coeffs.double().cpu(), None if knots is None else knots.double().cpu())
    z = z0.double().cpu().clone().requires_grad_(True)
    out = oracle_cde.cdeint(X, f64, z, t_out.double().cpu(), adjoint=True, method="rk4", options=dict(step_size=step))
    w = torch.ones_like(out) if loss_weight is None else loss_weight.double().cpu()
    (out * w).sum().backward()
    return out.detach(), z.grad, f64.linear.weight.grad, f64.linear.bias.grad


@pytest.mark.parametrize("variant,act", [("mfma", False), ("generic", False), ("generic", True), ("mfma", True)])
def test_cdeint_vs_float64_oracle_ragged_batch(native, variant, act):
    """B = 203 (not a multiple of the 32-series wave tile), 3 output times, fp32 kernels vs fp64 oracle."""
    B, L, C, H = 203, 24, 8, 32
    x = make_series(B, L, C, torch.float32, seed=21)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    func = LinearField(H, C, torch.float32, scale=0.25, tanh=act, seed=5)
    gen = torch.Generator().manual_seed(6)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 7.5, 23.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    ref_out, ref_gz, ref_gw, ref_gb = _oracle_solution(coeffs, None, func, z0, t_out, 1.0, lw)

    dfunc = LinearField(H, C, torch.float32, scale=0.25, tanh=act, seed=5).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, dfunc, z, t_out.to(DEV), method="rk4", options=dict(step_size=1.0), variant=variant)
    _close(out, ref_out, 1e-4, 1e-6)
    (out * lw.to(DEV)).sum().backward()
    _close(z.grad, ref_gz, 1e-3, 1e-5)
    _close(dfunc.linear.weight.grad, ref_gw, 1e-3, 1e-3 * ref_gw.abs().max().item())
    _close(dfunc.linear.bias.grad, ref_gb, 1e-3, 1e-3 * ref_gb.abs().max().item())


@pytest.mark.parametrize("H,C,degree", [(32, 8, 3), (32, 8, 1), (20, 5, 3)])
def test_tanh_field_forward_on_mfma_tiles(native, H, C, degree):
    """Linear -> tanh -> view(H, C) fields run the pre-activation tiling (cde_mfma.h: field_act16): forward solve
    vs the float64 oracle and vs the generic kernel, cubic and linear control, padded shapes."""
    B, L = 203, 24
    x = make_series(B, L, C, torch.float32, seed=31)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    func = LinearField(H, C, torch.float32, scale=0.5, tanh=True, seed=9)
    gen = torch.Generator().manual_seed(10)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 7.5, 23.])
    f64 = LinearField(H, C, torch.float64, scale=0.5, tanh=True, seed=9)
    f64.linear.weight.data.copy_(func.linear.weight.double()); f64.linear.bias.data.copy_(func.linear.bias.double())
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(coeffs.double())
    with torch.no_grad():
        ref = oracle_cde.cdeint(path64, f64, z0.double(), t_out.double(), adjoint=False, method="rk4",
                                options=dict(step_size=1.0))
    dfunc = LinearField(H, C, torch.float32, scale=0.5, tanh=True, seed=9).to(DEV)
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs.to(DEV))
    res = {}
    with torch.no_grad():
        for variant in ("mfma", "generic"):
            res[variant] = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0),
                                         variant=variant)
    _close(res["mfma"], ref, 1e-4, 5e-6)
    _close(res["mfma"], res["generic"], 1e-4, 5e-6)


@pytest.mark.parametrize("H,C,degree,act", [(32, 8, 3, False), (32, 8, 3, True), (16, 5, 1, True), (8, 3, 1, False)])
def test_gradient_wrt_control_coefficients(native, H, C, degree, act):
    """adjoint_params = func parameters + the coefficient tensor (reference README.md:251-270, solver.py:207-222):
    dL/dcoeffs from the fused adjoint against the float64 oracle, cubic and linear control, 3 output times."""
    B, L = 70, 12
    x = make_series(B, L, C, torch.float32, seed=81)
    base = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    gen = torch.Generator().manual_seed(82)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 4.5, 11.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    # oracle, float64
    f64 = LinearField(H, C, torch.float64, scale=0.4, tanh=act, seed=7)
    c64 = base.double().clone().requires_grad_(True)
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(c64)
    ref = oracle_cde.cdeint(path64, f64, z0.double(), t_out.double(), adjoint=True, method="rk4",
                            options=dict(step_size=1.0), adjoint_params=tuple(f64.parameters()) + (c64,))
    (ref * lw.double()).sum().backward()
    assert c64.grad is not None and c64.grad.abs().max() > 0
    # native
    dfunc = LinearField(H, C, torch.float32, scale=0.4, tanh=act, seed=7).to(DEV)
    coeffs = base.to(DEV).requires_grad_(True)
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs)
    out = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0),
                        adjoint_params=tuple(dfunc.parameters()) + (coeffs,))
    _close(out, ref, 1e-4, 5e-6)
    (out * lw.to(DEV)).sum().backward()
    assert coeffs.grad is not None and coeffs.grad.shape == coeffs.shape
    _close(coeffs.grad, c64.grad, 1e-3, 1e-3 * c64.grad.abs().max().item())
    _close(dfunc.linear.weight.grad, f64.linear.weight.grad, 1e-3, 1e-3 * f64.linear.weight.grad.abs().max().item())
    if degree == 3:
        assert torch.count_nonzero(coeffs.grad[..., :C]) == 0        # the derivative never reads the `a` block
    # without adjoint_params the reference only warns (solver.py:207-222) and leaves the control without gradient
    coeffs.grad = None
    with pytest.warns(UserWarning):
        out = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0))
    out.sum().backward()
    assert coeffs.grad is None


class _TwoLayerField(torch.nn.Module):
    """reference example/time_series_classification.py:20-51"""

    def __init__(self, H, C, width, dtype=torch.float32, seed=0, final_tanh=True):
        super().__init__()
        self.H, self.C, self.final_tanh = H, C, final_tanh
        gen = torch.Generator().manual_seed(seed)
        self.linear1 = torch.nn.Linear(H, width).to(dtype)
        self.linear2 = torch.nn.Linear(width, H * C).to(dtype)
        with torch.no_grad():
            for lin in (self.linear1, self.linear2):
                bound = 1 / lin.in_features ** 0.5
                lin.weight.copy_((torch.rand(lin.weight.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
                lin.bias.copy_((torch.rand(lin.bias.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)

    def forward(self, t, z):
        y = self.linear2(self.linear1(z).relu())
        if self.final_tanh:
            y = y.tanh()
        return y.view(*z.shape[:-1], self.H, self.C)


@pytest.mark.parametrize("H,C,width,degree,final_tanh", [(32, 8, 128, 3, True), (16, 4, 64, 1, True), (8, 3, 100, 3, False)])
def test_two_layer_field_forward_fused(native, H, C, width, degree, final_tanh):
    """Linear -> relu -> Linear -> tanh fields: the fused forward kernel (K2m) vs the float64 oracle and vs the
    step-wise path running the user module itself."""
    from torchcde_amd import fields
    B, L = 203, 24
    x = make_series(B, L, C, torch.float32, seed=61)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    func = _TwoLayerField(H, C, width, seed=3, final_tanh=final_tanh)
    f64 = _TwoLayerField(H, C, width, torch.float64, seed=3, final_tanh=final_tanh)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(62))
    t_out = torch.tensor([0., 7.5, 23.])
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(coeffs.double())
    with torch.no_grad():
        ref = oracle_cde.cdeint(path64, f64, z0.double(), t_out.double(), adjoint=False, method="rk4",
                                options=dict(step_size=1.0))
    dfunc = _TwoLayerField(H, C, width, seed=3, final_tanh=final_tanh).to(DEV)
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs.to(DEV))
    found, _ = fields.probe(dfunc, t_out[0].to(DEV), z0.to(DEV))
    assert found is not None and found.kind == "mlp2"
    with torch.no_grad():
        fused = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0))
        stepwise = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0),
                                 variant="generic")
    _close(fused, ref, 1e-4, 5e-6)
    _close(fused, stepwise, 1e-4, 5e-6)
    assert not torch.equal(fused, stepwise)          # two different code paths did run
    # the reference's default method: adaptive dopri5 (fused attempt kernel vs the host-driven controller vs float64)
    from torchcde_amd.cdeint import last_dopri5_stats
    kw = dict(method="dopri5", options=dict(jump_t=X.grid_points)) if degree == 1 else {}
    with torch.no_grad():
        last_dopri5_stats.clear()
        fused5 = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), **kw)
        assert last_dopri5_stats["n_accept"] > 0                       # the fused K4 loop ran
        stepwise5 = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), variant="generic", **kw)
        fine = oracle_cde.cdeint(path64, f64, z0.double(), t_out.double(), adjoint=False, method="rk4",
                                 options=dict(step_size=0.0625))
    # Two float32 controllers with independent rounding take different step sequences, so they agree with each other
    # only to the GLOBAL error of an rtol=1e-4 solve (a few 1e-3 of the state here: relu kinks, piecewise-cubic
    # control); measure both against a finely stepped float64 solution instead.
    scale = fine.abs().max().item()
    err_fused = (fused5.double().cpu() - fine).abs().max().item()
    err_step = (stepwise5.double().cpu() - fine).abs().max().item()
    assert err_fused <= 4 * err_step + 2e-3 * scale, (err_fused, err_step, scale)


@pytest.mark.parametrize("H,C,width,degree,final_tanh,chunk_bytes",
                         [(32, 8, 128, 3, True, None), (16, 4, 64, 1, True, 1), (8, 3, 100, 3, False, None)])
def test_two_layer_field_adjoint_fused(native, H, C, width, degree, final_tanh, chunk_bytes):
    """Training path of the example model: fused forward + continuous-adjoint sweep (K3m) + GEMM reduction against
    the float64 oracle's odeint_adjoint restatement.  3 output times (two reverse segments with re-seeding),
    ragged batch; `chunk_bytes=1` forces one sweep launch per step (state carried through HBM between launches)."""
    import importlib
    cdeint_mod = importlib.import_module("torchcde_amd.cdeint")      # the package attribute `cdeint` is the function
    B, L = 203, 24
    x = make_series(B, L, C, torch.float32, seed=71)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    gen = torch.Generator().manual_seed(72)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 7.5, 23.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    f64 = _TwoLayerField(H, C, width, torch.float64, seed=5, final_tanh=final_tanh)
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(coeffs.double())
    zr = z0.double().clone().requires_grad_(True)
    ref = oracle_cde.cdeint(path64, f64, zr, t_out.double(), adjoint=True, method="rk4", options=dict(step_size=1.0))
    (ref * lw.double()).sum().backward()
    # The relu makes the gradient discontinuous in z: float32 and float64 trajectories differ by ~1e-6, a few of the
    # 203 * 508 * 128 hidden units sit within that distance of zero and flip, and each flip moves the gradient by a
    # discrete amount.  The CPU float32 run of the same algorithm measures how large that effect is here.
    f32 = _TwoLayerField(H, C, width, torch.float32, seed=5, final_tanh=final_tanh)
    path32 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(coeffs)
    z32 = z0.clone().requires_grad_(True)
    (oracle_cde.cdeint(path32, f32, z32, t_out, adjoint=True, method="rk4", options=dict(step_size=1.0)) * lw).sum().backward()

    def bar(want, cpu32):          # rtol 1e-3 of the largest entry, or 4x what float32 costs on the CPU
        return max(1e-3 * want.abs().max().item(), 4 * (cpu32.double() - want).abs().max().item())

    dfunc = _TwoLayerField(H, C, width, seed=5, final_tanh=final_tanh).to(DEV)
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs.to(DEV))
    z = z0.to(DEV).requires_grad_(True)
    budget = cdeint_mod._MlpPlan.scratch_budget
    try:
        if chunk_bytes is not None:
            cdeint_mod._MlpPlan.scratch_budget = chunk_bytes
        out = native.cdeint(X, dfunc, z, t_out.to(DEV), method="rk4", options=dict(step_size=1.0))
        assert type(out.grad_fn).__name__ == "_FusedMlpRK4Backward"          # the fused path, not the step-wise one
        (out * lw.to(DEV)).sum().backward()
    finally:
        cdeint_mod._MlpPlan.scratch_budget = budget
    _close(out, ref, 1e-4, 5e-6)
    _close(z.grad, zr.grad, 1e-3, bar(zr.grad, z32.grad))
    for name in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"):
        layer, kind = name.split(".")
        got = getattr(getattr(dfunc, layer), kind).grad
        want = getattr(getattr(f64, layer), kind).grad
        cpu32 = getattr(getattr(f32, layer), kind).grad
        assert got.shape == want.shape
        _close(got, want, 1e-3, bar(want, cpu32))


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("H,C", [(16, 4), (32, 3), (5, 2), (24, 8)])
def test_mfma_kernels_on_zero_padded_shapes(native, H, C, act):
    """H <= 32, C <= 8 run on the MFMA tiles zero-padded (weight images, hidden units and channels outside the real
    shape are zeros that are never stored): forward, adjoint and dopri5 against the float64 oracle / generic kernel."""
    B, L = 75, 12
    x = make_series(B, L, C, torch.float32, seed=50 + H)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    func = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=H)
    gen = torch.Generator().manual_seed(H)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 4.5, 11.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    ref_out, ref_gz, ref_gw, ref_gb = _oracle_solution(coeffs, None, func, z0, t_out, 1.0, lw)
    dfunc = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=H).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, dfunc, z, t_out.to(DEV), method="rk4", options=dict(step_size=1.0), variant="mfma")
    _close(out, ref_out, 1e-4, 1e-6)
    (out * lw.to(DEV)).sum().backward()
    _close(z.grad, ref_gz, 1e-3, 1e-5)
    _close(dfunc.linear.weight.grad, ref_gw, 1e-3, 1e-3 * ref_gw.abs().max().item())
    _close(dfunc.linear.bias.grad, ref_gb, 1e-3, 1e-3 * ref_gb.abs().max().item())
    res = {}
    for variant in ("mfma", "generic"):
        with torch.no_grad():
            res[variant] = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="dopri5",
                                         options=dict(jump_t=X.grid_points), variant=variant)
    _close(res["mfma"], res["generic"], 3e-3, 3e-3 * res["generic"].abs().max().item())


def test_linear_control_path(native):
    """LinearInterpolation control (config 4's control type) through the same fused kernels."""
    B, L, C, H = 70, 20, 8, 32
    x = make_series(B, L, C, torch.float32, seed=2)
    func = LinearField(H, C, torch.float32, scale=0.25, seed=1)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(1))
    f64 = LinearField(H, C, torch.float64, scale=0.25, seed=1)
    Xo = oracle_interp.LinearPath(x.double())
    z = z0.double().requires_grad_(True)
    ref = oracle_cde.cdeint(Xo, f64, z, Xo.interval, adjoint=True, method="rk4", options=dict(step_size=0.5))
    ref.sum().backward()
    for variant in ("mfma", "generic"):
        dfunc = LinearField(H, C, torch.float32, scale=0.25, seed=1).to(DEV)
        X = native.LinearInterpolation(native.linear_interpolation_coeffs(x.to(DEV)))
        zd = z0.to(DEV).requires_grad_(True)
        out = native.cdeint(X, dfunc, zd, X.interval, method="rk4", options=dict(step_size=0.5), variant=variant)
        _close(out, ref, 1e-4, 1e-6)
        out.sum().backward()
        _close(zd.grad, z.grad, 1e-3, 1e-5)
        _close(dfunc.linear.weight.grad, f64.linear.weight.grad, 1e-3, 1e-3 * f64.linear.weight.grad.abs().max().item())


def test_float64_generic_kernel_vs_oracle_tight(native):
    B, L, C, H = 9, 11, 3, 5
    x = make_series(B, L, C, torch.float64, seed=8)
    knots = (torch.rand(L, dtype=torch.float64).cumsum(0) + 0.3)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x, knots)
    func = LinearField(H, C, torch.float64, scale=0.5, tanh=True, seed=4)
    z0 = torch.randn(B, H, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    t_out = torch.stack([knots[0], (knots[0] + knots[-1]) / 2, knots[-1]])
    ref_out, ref_gz, ref_gw, ref_gb = _oracle_solution(coeffs, knots, func, z0, t_out, 0.4)
    dfunc = LinearField(H, C, torch.float64, scale=0.5, tanh=True, seed=4).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV), knots.to(DEV))
    z = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, dfunc, z, t_out.to(DEV), method="rk4", options=dict(step_size=0.4))
    _close(out, ref_out, 1e-10, 1e-12)
    out.sum().backward()
    _close(z.grad, ref_gz, 1e-9, 1e-11)
    _close(dfunc.linear.weight.grad, ref_gw, 1e-9, 1e-10)
    _close(dfunc.linear.bias.grad, ref_gb, 1e-9, 1e-10)


def test_batch_dims_time_dtype_and_adjoint_params(native):
    """(2,3) batch dims, float64 output times with float32 state (reference test_cdeint.py:43), adjoint_params subset."""
    x = make_series(6, 10, 2, torch.float32, seed=12).view(2, 3, 10, 2)
    coeffs = native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV))
    X = native.CubicSpline(coeffs)
    func = LinearField(3, 2, scale=0.5, seed=2).to(DEV)
    z0 = torch.rand(2, 3, 3, device=DEV, requires_grad=True)
    t = torch.tensor([0.4, 3.3, 8.1], dtype=torch.float64, device=DEV)
    out = native.cdeint(X, func, z0, t, method="rk4", options=dict(step_size=1.0),
                        adjoint_params=(func.linear.weight,))
    assert out.shape == (2, 3, 3, 3)
    out.sum().backward()
    assert func.linear.weight.grad is not None and func.linear.bias.grad is None and z0.grad.shape == z0.shape
    fo = LinearField(3, 2, scale=0.5, seed=2)
    Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x))
    ref = oracle_cde.cdeint(Xo, fo, z0.detach().cpu(), t.cpu(), adjoint=False, method="rk4", options=dict(step_size=1.0))
    _close(out, ref, 1e-4, 1e-6)


def test_gradients_are_run_to_run_deterministic(native):
    """Parameter gradients are reduced in a fixed order (no atomics): two runs are bit-identical, which is
    what the reference's detach-trick test relies on (test/test_tricks.py:111-131)."""
    B, L = 1000, 32
    coeffs = native.hermite_cubic_coefficients_with_backward_differences(make_series(B, L, 8, seed=9).to(DEV))
    X = native.CubicSpline(coeffs)
    func = LinearField(32, 8, scale=0.25).to(DEV)
    z0 = torch.randn(B, 32, device=DEV)
    grads = []
    for _ in range(2):
        func.zero_grad()
        z = z0.clone().requires_grad_(True)
        native.cdeint(X, func, z, X.interval, method="rk4", options=dict(step_size=1.0))[:, -1].sum().backward()
        grads.append((z.grad.clone(), func.linear.weight.grad.clone(), func.linear.bias.grad.clone()))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_unsupported_requests_fail_loudly(native):
    coeffs = torch.randn(4, 5, 8, device=DEV)
    X = native.CubicSpline(coeffs)
    func = LinearField(3, 2).to(DEV)
    z0 = torch.randn(4, 3, device=DEV)
    with pytest.raises(NotImplementedError, match="bosh3"):
        native.cdeint(X, func, z0, X.interval, method="bosh3")
    with pytest.raises(ValueError, match="same number of batch dimensions as z0"):
        native.cdeint(X, func, torch.randn(5, 3, device=DEV), X.interval, method="rk4")
    with pytest.raises(ValueError, match="same number of input channels"):
        native.cdeint(X, LinearField(3, 4).to(DEV), z0, X.interval, method="rk4")

    with pytest.raises(NotImplementedError, match="prod"):
        class P:
            def prod(self, t, z, dXdt):
                return -z
        native.cdeint(X, P(), z0, X.interval, method="rk4", adjoint_params=())


# =========================================================================================== step-wise path
class _Mlp(torch.nn.Module):
    """The vector field of reference example/time_series_classification.py:30-51."""

    def __init__(self, C, H, width, dtype, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.C, self.H = C, H
        self.linear1 = torch.nn.Linear(H, width).to(dtype)
        self.linear2 = torch.nn.Linear(width, C * H).to(dtype)

    def forward(self, t, z):
        return self.linear2(self.linear1(z).relu()).tanh().view(z.size(0), self.H, self.C)


@pytest.mark.parametrize("method,options,adjoint", [("rk4", dict(step_size=0.5), True), ("rk4", dict(step_size=0.5), False),
                                                   ("dopri5", None, True), ("midpoint", dict(step_size=0.25), False)])
def test_stepwise_path_arbitrary_func_vs_oracle(native, method, options, adjoint):
    """Arbitrary nn.Module vector fields (here the example's 2-layer MLP) run step-wise on the GPU with the native
    control-derivative and contraction kernels; float64 so the comparison with the oracle is tight."""
    B, L, C, H = 6, (6 if method == "dopri5" else 10), 3, 8
    dtype = torch.float64
    x = make_series(B, L, C, dtype, seed=17)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    z0 = torch.randn(B, H, dtype=dtype, generator=torch.Generator().manual_seed(17))
    kw = dict(method=method, adjoint=adjoint)
    if options is not None:
        kw["options"] = options
    if method == "dopri5":
        kw.update(rtol=1e-5, atol=1e-7)
    fo = _Mlp(C, H, 16, dtype, seed=3)
    zo = z0.clone().requires_grad_(True)
    Xo = oracle_interp.CubicPath(coeffs)
    ref = oracle_cde.cdeint(Xo, fo, zo, Xo.interval, **kw)
    ref[:, -1].pow(2).sum().backward()

    fd = _Mlp(C, H, 16, dtype, seed=3).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    zd = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, fd, zd, X.interval, **kw)
    assert out.shape == ref.shape
    out[:, -1].pow(2).sum().backward()
    # adaptive: two tolerance-level solutions (forward AND adjoint solve) of a field with ReLU kinks whose step
    # sequences drift apart on round-off -> compared at 100x the requested tolerance, see the K4 tests
    tight = method != "dopri5"
    _close(out, ref, 1e-9 if tight else 1e-3, 1e-11 if tight else 1e-4)
    _close(zd.grad, zo.grad, 1e-8 if tight else 1e-2, 1e-10 if tight else 1e-3)
    for pd, po in zip(fd.parameters(), fo.parameters()):
        _close(pd.grad, po.grad, 1e-8 if tight else 1e-2, (1e-10 if tight else 1e-3) * max(1.0, po.grad.abs().max().item()))


def test_recognised_field_gradients_through_dopri5_and_backprop_mode(native):
    """Requests the fused kernels do not cover for the affine family (gradients through dopri5, adjoint=False
    backprop) take the step-wise path instead of failing."""
    B, L, C, H = 5, 8, 8, 32
    x = make_series(B, L, C, torch.float64, seed=23)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    z0 = torch.randn(B, H, dtype=torch.float64, generator=torch.Generator().manual_seed(23))
    for kw in (dict(method="rk4", options=dict(step_size=1.0), adjoint=False),
               dict(method="dopri5", rtol=1e-5, atol=1e-7, adjoint=True)):
        fo = LinearField(H, C, torch.float64, scale=0.25, seed=6)
        zo = z0.clone().requires_grad_(True)
        Xo = oracle_interp.CubicPath(coeffs)
        ref = oracle_cde.cdeint(Xo, fo, zo, Xo.interval, **kw)
        ref[:, -1].sum().backward()
        fd = LinearField(H, C, torch.float64, scale=0.25, seed=6).to(DEV)
        zd = z0.to(DEV).requires_grad_(True)
        X = native.CubicSpline(coeffs.to(DEV))
        out = native.cdeint(X, fd, zd, X.interval, **kw)
        out[:, -1].sum().backward()
        tol = 1e-8 if kw["method"] == "rk4" else 1e-2     # adaptive forward + adaptive adjoint: tolerance-level
        _close(out, ref, tol, tol * 1e-2)
        _close(zd.grad, zo.grad, tol, tol * 1e-2)
        # parameter gradients of an adaptive adjoint solve carry ~100 steps x rtol of drift: bar = 1 % of the largest entry
        _close(fd.linear.weight.grad, fo.linear.weight.grad, tol, tol * fo.linear.weight.grad.abs().max().item())


# =========================================================================================== dopri5 (K4)
def test_dopri5_default_method_vs_reference_golden(native, golden_cde):
    """cdeint with no method = torchdiffeq's dopri5 at rtol 1e-4 / atol 1e-6 (solver.py:195-198); README toy included.
    Both sides are adaptive solutions accurate to about the tolerance, so they are compared at 10x that."""
    ran = 0
    for case in golden_cde:
        if case["method"] not in (None, "dopri5"):
            continue
        func = golden_field(case).to(DEV)
        X = native.CubicSpline(case["coeffs"].to(DEV))
        kw = {} if case["method"] is None else dict(method=case["method"])
        with torch.no_grad():
            out = native.cdeint(X, func, case["z0"].to(DEV), case["t_out"].to(DEV), **kw)
        assert out.shape == case["out_direct"].shape
        # two adaptive solutions at rtol 1e-4 whose step sequences may differ, on an expanding system (the README
        # toy grows to |z| ~ 20, so early differences are amplified): agreement at 50x the tolerance, measured
        # against the size of the trajectory
        ref = case["out_direct"]
        _close(out, ref, 5e-3, 5e-3 * ref.abs().max().item())
        from torchcde_amd.cdeint import last_dopri5_stats
        assert last_dopri5_stats["n_accept"] > 0
        ran += 1
    assert ran == 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_dopri5_controller_matches_oracle_step_for_step(native, dtype):
    """Config-4 shaped problem in miniature: LinearInterpolation control, jump_t at the knots, several output
    times.  The batch-global controller must take the oracle's accept / reject sequence (same counts) and land on
    the same trajectory far inside the solver tolerance."""
    from oracle import odeint as oracle_ode
    B, L, C, H = 37, 14, 8, 32
    x = make_series(B, L, C, dtype, seed=31)
    func = LinearField(H, C, dtype, scale=0.25, seed=3)
    z0 = torch.randn(B, H, dtype=dtype, generator=torch.Generator().manual_seed(2))
    Xo = oracle_interp.LinearPath(x)
    t_out = torch.tensor([0., 2.25, 6.5, 13.], dtype=dtype)
    field = oracle_ode._Field(oracle_cde.ControlledField(Xo, func))
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float64 else (1e-4, 1e-6)     # float32: the reference's defaults
    solver = oracle_ode._Dopri5(field, z0, rtol, atol, oracle_ode._rms, jump_t=Xo.grid_points)
    with torch.no_grad():
        ref = solver.integrate(t_out).permute(1, 0, 2)
    dfunc = LinearField(H, C, dtype, scale=0.25, seed=3).to(DEV)
    X = native.LinearInterpolation(x.to(DEV))
    with torch.no_grad():
        out = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="dopri5", rtol=rtol, atol=atol,
                            options=dict(jump_t=X.grid_points))
    from torchcde_amd.cdeint import last_dopri5_stats
    got = (last_dopri5_stats["n_accept"], last_dopri5_stats["n_reject"])
    if dtype == torch.float64:
        assert got == (solver.n_accept, solver.n_reject)          # identical accept / reject sequence
        # same steps, but the embedded error estimate is a cancelling sum: round-off level differences in f
        # (GEMM summation order) move each dt by ~1e-8 relative, hence agreement far below rtol, not bitwise
        _close(out, ref, 1e-7, 1e-8)
    else:
        # float32: the error ratio is reduced in a different order, decisions with ratio ~ 1 may flip and the
        # sequences drift apart; both remain valid solutions at the requested tolerance
        assert abs(got[0] - solver.n_accept) <= 0.15 * solver.n_accept, (got, solver.n_accept, solver.n_reject)
        _close(out, ref, 2e-3, 2e-3 * ref.abs().max().item())


@pytest.mark.parametrize("act", [False, True])
def test_dopri5_mfma_kernel_equals_generic_kernel(native, act):
    """Same controller, same state layout: the MFMA attempt kernel must take the generic kernel's step sequence."""
    from torchcde_amd.cdeint import last_dopri5_stats
    B, L, C, H = 300, 20, 8, 32                                 # ragged: 300 = 2*128 + 44
    x = make_series(B, L, C, seed=41).to(DEV)
    func = LinearField(H, C, scale=0.25, tanh=act, seed=4).to(DEV)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(4)).to(DEV)
    t_out = torch.tensor([0., 3.3, 19.], device=DEV)
    res = {}
    for control in ("linear", "cubic"):
        X = (native.LinearInterpolation(x) if control == "linear"
             else native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x)))
        for variant in ("mfma", "generic"):
            with torch.no_grad():
                out = native.cdeint(X, func, z0, t_out, method="dopri5", options=dict(jump_t=X.grid_points),
                                    variant=variant)
            res[variant] = (out, last_dopri5_stats["n_accept"], last_dopri5_stats["n_reject"])
        assert abs(res["mfma"][1] - res["generic"][1]) <= 2
        _close(res["mfma"][0], res["generic"][0], 1e-3, 1e-4)


def test_dopri5_cubic_control_without_jumps(native):
    B, L, C, H = 10, 9, 3, 5
    x = make_series(B, L, C, torch.float64, seed=5)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    func = LinearField(H, C, torch.float64, scale=0.5, tanh=True, seed=8)
    z0 = torch.randn(B, H, dtype=torch.float64, generator=torch.Generator().manual_seed(8))
    Xo = oracle_interp.CubicPath(coeffs)
    ref = oracle_cde.cdeint(Xo, func, z0, Xo.interval, adjoint=False, method="dopri5", rtol=1e-8, atol=1e-10)
    dfunc = LinearField(H, C, torch.float64, scale=0.5, tanh=True, seed=8).to(DEV)
    X = native.CubicSpline(coeffs.to(DEV))
    with torch.no_grad():
        out = native.cdeint(X, dfunc, z0.to(DEV), X.interval, method="dopri5", rtol=1e-8, atol=1e-10)
    _close(out, ref, 1e-5, 2e-6)       # stiff-ish tanh field: attempt sequences diverge after a near-tie (see above)


# =========================================================================================== full size
def test_full_size_properties_config2_config3(native):
    """BASELINE configs 2/3 (B=32768, L=128, C=8, H=32, fp32, RK4 step 1): size-independent properties.
      * series independence: reversing the batch reverses the result bit-for-bit
      * affine structure: z_T is an affine map of z0 for the affine field, RK4 preserves that
      * MFMA kernel == generic kernel == float64 oracle on a sample of series (trajectory + grad_z0)
      * parameter gradients of the two kernels agree"""
    B, L, C, H = 32768, 128, 8, 32
    x = make_series(B, L, C, seed=0).to(DEV)
    coeffs = native.hermite_cubic_coefficients_with_backward_differences(x)
    X = native.CubicSpline(coeffs)
    func = LinearField(H, C, scale=0.25, seed=0).to(DEV)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(0)).to(DEV)
    kw = dict(method="rk4", options=dict(step_size=1.0))

    z = z0.clone().requires_grad_(True)
    out = native.cdeint(X, func, z, X.interval, **kw)
    assert out.shape == (B, 2, H) and torch.isfinite(out).all()
    assert torch.equal(out[:, 0], z0)
    out[:, -1].sum().backward()
    gW, gb, gz = func.linear.weight.grad.clone(), func.linear.bias.grad.clone(), z.grad.clone()

    # series independence (bit-exact)
    Xr = native.CubicSpline(coeffs.flip(0).contiguous())
    out_r = native.cdeint(Xr, func, z0.flip(0).contiguous(), X.interval, **kw)
    assert torch.equal(out_r.flip(0), out.detach())

    # affine in z0: z(a) + z(b) - 2 z((a+b)/2) = 0
    zb = torch.randn(B, H, generator=torch.Generator().manual_seed(1)).to(DEV)
    mid = native.cdeint(X, func, 0.5 * (z0 + zb), X.interval, **kw)[:, -1]
    other = native.cdeint(X, func, zb, X.interval, **kw)[:, -1]
    resid = (out.detach()[:, -1] + other - 2 * mid).abs().max().item()
    assert resid < 1e-4 * max(1.0, out.detach().abs().max().item()), resid

    # sample vs float64 oracle and vs the generic kernel
    sample = torch.arange(0, B, 2048, device=DEV)
    Xs = native.CubicSpline(coeffs[sample].contiguous())
    zs = z0[sample].clone().requires_grad_(True)
    func_g = LinearField(H, C, scale=0.25, seed=0).to(DEV)
    out_g = native.cdeint(Xs, func_g, zs, X.interval, variant="generic", **kw)
    _close(out.detach()[sample], out_g, 1e-4, 1e-5)
    out_g[:, -1].sum().backward()
    _close(gz[sample], zs.grad, 1e-3, 1e-5)
    lw = torch.cat([torch.zeros(len(sample), 1, H), torch.ones(len(sample), 1, H)], 1)
    ref_out, ref_gz, _, _ = _oracle_solution(coeffs[sample].cpu(), None, LinearField(H, C, scale=0.25, seed=0),
                                             z0[sample].cpu(), X.interval.cpu(), 1.0, lw)
    _close(out.detach()[sample], ref_out, 1e-4, 1e-5)
    _close(gz[sample], ref_gz, 1e-3, 1e-5)
    # calibration: how far is the CPU float32 path (the reference's own arithmetic) from float64?
    cpu32 = oracle_cde.cdeint(oracle_interp.CubicPath(coeffs[sample].cpu()), LinearField(H, C, scale=0.25, seed=0),
                              z0[sample].cpu(), X.interval.cpu(), adjoint=False, method="rk4",
                              options=dict(step_size=1.0))
    err_cpu32 = (cpu32.double() - ref_out).abs().max().item()
    err_kernel = (out.detach()[sample].double().cpu() - ref_out).abs().max().item()
    assert err_kernel <= 4 * err_cpu32 + 1e-6, (err_kernel, err_cpu32)

    # parameter gradients: MFMA kernel vs generic kernel on a 4096-series slab
    slab = slice(0, 4096)
    res = []
    for variant in ("mfma", "generic"):
        f = LinearField(H, C, scale=0.25, seed=0).to(DEV)
        zz = z0[slab].clone().requires_grad_(True)
        o = native.cdeint(native.CubicSpline(coeffs[slab].contiguous()), f, zz, X.interval, variant=variant, **kw)
        o[:, -1].sum().backward()
        res.append((f.linear.weight.grad.clone(), f.linear.bias.grad.clone()))
    _close(res[0][0], res[1][0], 1e-3, 1e-4 * res[1][0].abs().max().item())
    _close(res[0][1], res[1][1], 1e-3, 1e-4 * res[1][1].abs().max().item())
    assert torch.isfinite(gW).all() and torch.isfinite(gb).all()
