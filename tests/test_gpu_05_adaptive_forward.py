"""GPU parity tests through the C ABI -- Row a11 (dopri5): K4, the adaptive forward solve and its step controller.

Tolerances and helpers: tests/gpu_common.py.  Collection order is the file order (01 first): the tests with the least driver history run first, so a failure elsewhere cannot hide them.
"""
import os

import pytest
import torch

from gpu_common import (oracle_cde, oracle_interp, LinearField, golden_field, make_series, DEV, _close)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,C,act,degree", [(64, 8, True, 3), (40, 6, False, 1), (32, 16, False, 3), (12, 11, True, 1)])
def test_wide_dopri5_attempt_kernel(native, H, C, act, degree):
    """The reference's default solver on the wide shapes, no gradients: the wide attempt kernel (K4 on the tiles of
    rk4_wide.hip) against the generic attempt kernel and a finely stepped float64 solution.  Three output times (dense
    output inside accepted steps), ragged batch over several tiles per workgroup, jump_t on the knots for the
    piecewise-linear control (steps clipped onto jump times, f re-evaluated after them)."""
    from torchcde_amd.cdeint import last_dopri5_stats
    B, L = 4500, 14                       # 282 tiles: more than the 256 workgroups of the 8-wave shape
    x = make_series(B, L, C, torch.float32, seed=170 + H)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs.to(DEV))
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(H)).to(DEV)
    t_out = torch.tensor([0., 5.3, 13.]).to(DEV)
    kw = dict(options=dict(jump_t=X.grid_points)) if degree == 1 else {}
    func = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=H).to(DEV)
    res = {}
    with torch.no_grad():
        for variant in ("auto", "generic"):
            last_dopri5_stats.clear()
            res[variant] = native.cdeint(X, func, z0, t_out, variant=variant, **kw)
            assert last_dopri5_stats["n_accept"] > 0
            res[variant + "_steps"] = last_dopri5_stats["n_accept"]
    assert not torch.equal(res["auto"], res["generic"])
    sample = torch.arange(0, B, 300)
    f64 = LinearField(H, C, torch.float64, scale=0.3, tanh=act, seed=H)
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(coeffs[sample].double())
    with torch.no_grad():
        fine = oracle_cde.cdeint(path64, f64, z0[sample.to(DEV)].double().cpu(), t_out.double().cpu(), adjoint=False,
                                 method="rk4", options=dict(step_size=0.03125))
    scale = fine.abs().max().item()
    err_wide = (res["auto"][sample.to(DEV)].double().cpu() - fine).abs().max().item()
    err_generic = (res["generic"][sample.to(DEV)].double().cpu() - fine).abs().max().item()
    assert err_wide <= 4 * err_generic + 2e-3 * scale, (err_wide, err_generic, scale)
    assert abs(res["auto_steps"] - res["generic_steps"]) <= max(3, res["generic_steps"] // 10)


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
