"""GPU parity tests through the C ABI -- Row f3: gradients w.r.t. the control's coefficients, the output times and the knot times; stacked CDEs.

Tolerances and helpers: tests/gpu_common.py.  Collection order is the file order (01 first): the tests with the least driver history run first, so a failure elsewhere cannot hide them.
"""
import os

import pytest
import torch

from gpu_common import (_expect_dispatch, oracle_cde, oracle_interp, LinearField, _TwoLayerField, make_series, DEV, _close, _TricksFunc, _time_grad_case)

pytestmark = pytest.mark.gpu


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


def test_data_gradient_through_fit_and_solve(native):
    """End to end: dL/dx through hermite fit -> CubicSpline -> cdeint(adjoint_params=(..., coeffs)), the chain a
    learned embedding in front of the interpolation (or a stacked CDE, README.md:251-270) needs."""
    B, L, C, H = 40, 10, 6, 24
    x = make_series(B, L, C, torch.float32, seed=95)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(96))
    f64 = LinearField(H, C, torch.float64, scale=0.4, tanh=True, seed=9)
    x64 = x.double().requires_grad_(True)
    c64 = oracle_interp.hermite_bdiff_coeffs(x64)
    ref = oracle_cde.cdeint(oracle_interp.CubicPath(c64), f64, z0.double(), torch.tensor([0., 9.], dtype=torch.float64),
                            adjoint=True, method="rk4", options=dict(step_size=1.0),
                            adjoint_params=tuple(f64.parameters()) + (c64,))
    ref[:, -1].square().sum().backward()
    dfunc = LinearField(H, C, torch.float32, scale=0.4, tanh=True, seed=9).to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    coeffs = native.hermite_cubic_coefficients_with_backward_differences(xd)
    X = native.CubicSpline(coeffs)
    out = native.cdeint(X, dfunc, z0.to(DEV), X.interval, method="rk4", options=dict(step_size=1.0),
                        adjoint_params=tuple(dfunc.parameters()) + (coeffs,))
    out[:, -1].square().sum().backward()
    _close(xd.grad, x64.grad, 1e-3, 1e-3 * x64.grad.abs().max().item())


@pytest.mark.parametrize("H,C,width,degree", [(32, 8, 128, 3), (12, 5, 40, 1),
                                              (16, 14, 100, 3), (16, 16, 64, 1)])    # (the 16 x 16 tile layout: config 5's shape)
def test_two_layer_field_gradient_wrt_control_coefficients(native, H, C, width, degree):
    """adjoint_params = the four layer parameters + the coefficient tensor, two-layer field: dL/dcoeffs accumulated by
    the K3m sweep (several chunk launches: state and partial row gradients carried across them) vs the float64 oracle."""
    import importlib
    cdeint_mod = importlib.import_module("torchcde_amd.cdeint")
    B, L = 70, 12
    x = make_series(B, L, C, torch.float32, seed=101)
    base = oracle_interp.hermite_bdiff_coeffs(x) if degree == 3 else x
    gen = torch.Generator().manual_seed(102)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 4.5, 11.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    f64 = _TwoLayerField(H, C, width, torch.float64, seed=6)
    c64 = base.double().clone().requires_grad_(True)
    path64 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(c64)
    ref = oracle_cde.cdeint(path64, f64, z0.double(), t_out.double(), adjoint=True, method="rk4",
                            options=dict(step_size=1.0), adjoint_params=tuple(f64.parameters()) + (c64,))
    (ref * lw.double()).sum().backward()
    f32 = _TwoLayerField(H, C, width, torch.float32, seed=6)            # CPU float32: size of the relu-kink effect
    c32 = base.clone().requires_grad_(True)
    path32 = (oracle_interp.CubicPath if degree == 3 else oracle_interp.LinearPath)(c32)
    (oracle_cde.cdeint(path32, f32, z0, t_out, adjoint=True, method="rk4", options=dict(step_size=1.0),
                       adjoint_params=tuple(f32.parameters()) + (c32,)) * lw).sum().backward()

    dfunc = _TwoLayerField(H, C, width, seed=6).to(DEV)
    coeffs = base.to(DEV).requires_grad_(True)
    X = (native.CubicSpline if degree == 3 else native.LinearInterpolation)(coeffs)
    budget = cdeint_mod._MlpPlan.scratch_budget
    try:
        cdeint_mod._MlpPlan.scratch_budget = 3 * 4 * B * 552 * 4        # three RK steps per sweep launch
        out = native.cdeint(X, dfunc, z0.to(DEV), t_out.to(DEV), method="rk4", options=dict(step_size=1.0),
                            adjoint_params=tuple(dfunc.parameters()) + (coeffs,))
        _expect_dispatch("two_layer_rk4_control", out)
        (out * lw.to(DEV)).sum().backward()
    finally:
        cdeint_mod._MlpPlan.scratch_budget = budget
    want = c64.grad
    bar = max(1e-3 * want.abs().max().item(), 4 * (c32.grad.double() - want).abs().max().item())
    assert coeffs.grad is not None and coeffs.grad.shape == coeffs.shape
    _close(coeffs.grad, want, 1e-3, bar)
    w2 = f64.linear2.weight.grad
    _close(dfunc.linear2.weight.grad, w2, 1e-3, max(1e-3 * w2.abs().max().item(),
                                                    4 * (f32.linear2.weight.grad.double() - w2).abs().max().item()))


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
@pytest.mark.parametrize("adjoint", [True, False])
def test_gradients_reach_everything_like_reference_test_grad_paths(native, method, adjoint):
    """Reference test/test_tricks.py:21-49 on the native path: the SAME `t` goes into natural_cubic_coeffs and into
    CubicSpline, the raw path, z0, the field's parameter and the output times all require gradients, with and without
    the adjoint method (adjoint_params = parameters + (coeffs, t), as the reference passes them).  The reference only
    asserts that every gradient exists; here they are also compared with autograd through the float64 oracle."""
    dtype = torch.float64
    gen = torch.Generator().manual_seed(17)
    path0 = torch.rand(1, 10, 3, generator=gen, dtype=dtype)
    z00 = torch.rand(1, 3, generator=gen, dtype=dtype)

    def run(lib_interp, spline, solve, dev):
        t = torch.linspace(0, 9, 10, dtype=dtype, device=dev).requires_grad_(True)
        path = path0.to(dev).clone().requires_grad_(True)
        coeffs = lib_interp(path, t)
        X = spline(coeffs, t)
        z0 = z00.to(dev).clone().requires_grad_(True)
        func = _TricksFunc(3, 3, dtype).to(dev)
        t_ = torch.tensor([0., 9.], dtype=dtype, device=dev, requires_grad=True)
        kwargs = dict(adjoint_params=tuple(func.parameters()) + (coeffs, t)) if adjoint else {}
        z = solve(X, func, z0, t_, adjoint=adjoint, method=method, rtol=1e-8, atol=1e-10, **kwargs)
        assert z.shape == (1, 2, 3)
        for leaf in (t, path, z0, func.variable, t_):
            assert leaf.grad is None
        z[:, 1].sum().backward()
        grads = (t.grad, path.grad, z0.grad, func.variable.grad, t_.grad)
        assert all(isinstance(g, torch.Tensor) for g in grads)
        return [g.detach().cpu() for g in grads]

    got = run(native.natural_cubic_coeffs, native.CubicSpline, native.cdeint, DEV)
    assert all(bool(torch.isfinite(g).all()) for g in got)
    if method == "dopri5" and not adjoint:
        return        # autograd through an adaptive solver: the gradient of one particular step sequence, not comparable
    want = run(oracle_interp.natural_cubic_coeffs, oracle_interp.CubicPath, oracle_cde.cdeint, "cpu")
    tol = 1e-9 if method == "rk4" else 1e-6           # two adaptive solves differ at the level of their tolerance
    for name, a, b in zip(("t", "path", "z0", "variable", "t_"), got, want):
        assert torch.allclose(a, b, rtol=tol, atol=tol * max(1.0, b.abs().max().item())), (name, a, b)


def test_control_gradients_through_the_stepwise_path(native):
    """Gradients w.r.t. the coefficients for an ARBITRARY func (reference test/test_tricks.py:21-106 style): the path
    evaluation is differentiable (K1b backward), so both adjoint=True with adjoint_params=(..., coeffs) and
    adjoint=False backprop reach the data through fit -> spline -> solve.  Checked against the float64 oracle."""
    B, L, C, H = 6, 9, 2, 3

    class Sigmoid(torch.nn.Module):                       # the reference's test func: sigmoid(z)[..., None] + variable
        def __init__(self, dtype):
            super().__init__()
            self.variable = torch.nn.Parameter(torch.linspace(0.1, 0.7, C, dtype=dtype).view(1, 1, C))

        def forward(self, t, z):
            return z.sigmoid().unsqueeze(-1) + self.variable

    x = make_series(B, L, C, torch.float64, seed=161)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(162), dtype=torch.float64)
    t = torch.tensor([0., 3., 8.], dtype=torch.float64)
    kw = dict(method="rk4", options=dict(step_size=1.0))
    for adjoint in (True, False):
        f64 = Sigmoid(torch.float64)
        x64 = x.clone().requires_grad_(True)
        c64 = oracle_interp.hermite_bdiff_coeffs(x64)
        extra = dict(adjoint_params=tuple(f64.parameters()) + (c64,)) if adjoint else {}
        ref = oracle_cde.cdeint(oracle_interp.CubicPath(c64), f64, z0, t, adjoint=adjoint, **kw, **extra)
        ref.square().sum().backward()
        fd = Sigmoid(torch.float64).to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        coeffs = native.hermite_cubic_coefficients_with_backward_differences(xd)
        extra = dict(adjoint_params=tuple(fd.parameters()) + (coeffs,)) if adjoint else {}
        out = native.cdeint(native.CubicSpline(coeffs), fd, z0.to(DEV), t.to(DEV), adjoint=adjoint, **kw, **extra)
        _close(out, ref, 1e-9, 1e-11)
        out.square().sum().backward()
        _close(xd.grad, x64.grad, 1e-7, 1e-9 * x64.grad.abs().max().item())
        _close(fd.variable.grad, f64.variable.grad, 1e-7, 1e-9)
    # evaluate / derivative themselves are differentiable w.r.t. the coefficients
    cd = oracle_interp.hermite_bdiff_coeffs(x).to(DEV).requires_grad_(True)
    co = oracle_interp.hermite_bdiff_coeffs(x).clone().requires_grad_(True)
    tq = torch.tensor([0.3, 2.0, 2.7, 7.9], dtype=torch.float64)
    (native.CubicSpline(cd).evaluate(tq.to(DEV)).square().sum() + native.CubicSpline(cd).derivative(tq.to(DEV)).sum()).backward()
    (oracle_interp.CubicPath(co).evaluate(tq).square().sum() + oracle_interp.CubicPath(co).derivative(tq).sum()).backward()
    _close(cd.grad, co.grad, 1e-12, 1e-13)
    ld = x.to(DEV).clone().requires_grad_(True)
    lo = x.clone().requires_grad_(True)
    (native.LinearInterpolation(ld).evaluate(tq.to(DEV)).square().sum() + native.LinearInterpolation(ld).derivative(tq.to(DEV)).sum()).backward()
    (oracle_interp.LinearPath(lo).evaluate(tq).square().sum() + oracle_interp.LinearPath(lo).derivative(tq).sum()).backward()
    _close(ld.grad, lo.grad, 1e-12, 1e-13)


@pytest.mark.parametrize("act", [False, True])
def test_gradients_wrt_output_times_and_knot_times_fused(native, act):
    """reference test/test_tricks.py:21-49 asks for gradients w.r.t. the output times `t_` and (through adjoint_params)
    the control's knot times.  Fused rk4 path, cubic control, float32 kernels against the float64 oracle's
    odeint_adjoint (which restates torchdiffeq's time_vjps)."""
    x, knots, coeffs, z0, t_out, lw = _time_grad_case(torch.float32)
    H, C = 32, 8
    f64 = LinearField(H, C, torch.float64, scale=0.3, tanh=act, seed=2)
    kn = knots.double().requires_grad_(True)
    Xo = oracle_interp.CubicPath(coeffs.double(), kn)
    to = t_out.double().requires_grad_(True)
    zo = z0.double().requires_grad_(True)
    ref = oracle_cde.cdeint(Xo, f64, zo, to, adjoint=True, method="rk4", options=dict(step_size=0.25),
                            adjoint_params=tuple(f64.parameters()) + (kn,))
    (ref * lw.double()).sum().backward()

    func = LinearField(H, C, scale=0.3, tanh=act, seed=2).to(DEV)
    kd = knots.to(DEV).requires_grad_(True)
    X = native.CubicSpline(coeffs.to(DEV), kd)
    td = t_out.to(DEV).requires_grad_(True)
    zd = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, func, zd, td, method="rk4", options=dict(step_size=0.25),
                        adjoint_params=tuple(func.parameters()) + (kd,))
    _close(out, ref, 1e-4, 1e-5)
    (out * lw.to(DEV)).sum().backward()
    scale = to.grad.abs().max().item()
    _close(td.grad, to.grad, 2e-3, 2e-3 * scale)
    _close(kd.grad, kn.grad, 2e-3, 2e-3 * kn.grad.abs().max().item())
    _close(zd.grad, zo.grad, 1e-3, 1e-5)
    _close(func.linear.weight.grad, f64.linear.weight.grad, 1e-3, 1e-3 * f64.linear.weight.grad.abs().max().item())


def test_gradients_wrt_knot_times_of_a_linear_control_fused(native, variant="auto", dtype=torch.float32):
    """The knot times of a piecewise-linear control through adjoint_params on the fused rk4 path: the slopes
    (x_{j+1} - x_j) / (t_{j+1} - t_j) of interpolation_linear.py:189 depend on them; the gradient is recovered from the
    knot-value gradient of the adjoint sweep.  Against the float64 oracle's odeint_adjoint; the output times and the
    knot values require gradients in the same call."""
    x, knots, _, z0, t_out, lw = _time_grad_case(torch.float32)
    H, C = 32, 8
    f64 = LinearField(H, C, torch.float64, scale=0.3, seed=2)
    kn = knots.double().requires_grad_(True)
    xo = x.double().requires_grad_(True)
    to = t_out.double().requires_grad_(True)
    Xo = oracle_interp.LinearPath(xo, kn)
    ref = oracle_cde.cdeint(Xo, f64, z0.double(), to, adjoint=True, method="rk4", options=dict(step_size=0.25),
                            adjoint_params=tuple(f64.parameters()) + (xo, kn))
    (ref * lw.double()).sum().backward()
    # the identity the native path uses, in float64 on the oracle's own gradients: dL/dh_j = cumsum_j(dL/dx) . slope_j
    slopes = ((xo[:, 1:] - xo[:, :-1]) / (kn[1:] - kn[:-1]).unsqueeze(-1)).detach()
    dh = (xo.grad.cumsum(1)[:, :-1] * slopes).sum((0, 2))
    zero = dh.new_zeros(1)
    _close(torch.cat([zero, dh]) - torch.cat([dh, zero]), kn.grad, 1e-9, 1e-10 * kn.grad.abs().max().item())
    func = LinearField(H, C, dtype, scale=0.3, seed=2).to(DEV)
    kd = knots.to(DEV, dtype).requires_grad_(True)
    xd = x.to(DEV, dtype).requires_grad_(True)
    td = t_out.to(DEV, dtype).requires_grad_(True)
    X = native.LinearInterpolation(xd, kd)
    out = native.cdeint(X, func, z0.to(DEV, dtype), td, method="rk4", options=dict(step_size=0.25), variant=variant,
                        adjoint_params=tuple(func.parameters()) + (xd, kd))
    _expect_dispatch("affine_rk4_control", out)
    (out * lw.to(DEV, dtype)).sum().backward()
    tol = 2e-3 if dtype == torch.float32 else 1e-9
    _close(kd.grad, kn.grad, tol, tol * kn.grad.abs().max().item())
    _close(xd.grad, xo.grad, tol, tol * xo.grad.abs().max().item())
    _close(td.grad, to.grad, tol, tol * to.grad.abs().max().item())


def test_gradients_wrt_output_times_linear_control_and_stepwise(native):
    """Output-time gradients with a piecewise-linear control (no d2X/dt2 term) on the fused path, and for an arbitrary
    func (step-wise path, float64) -- both against the oracle; plus the reference's detach trick
    (test/test_tricks.py:111-131): parameter gradients are bitwise the same whether or not `t` requires grad."""
    x, knots, _, z0, t_out, lw = _time_grad_case(torch.float32)
    H, C = 32, 8
    f64 = LinearField(H, C, torch.float64, scale=0.3, seed=2)
    Xo = oracle_interp.LinearPath(x.double(), knots.double())
    to = t_out.double().requires_grad_(True)
    ref = oracle_cde.cdeint(Xo, f64, z0.double(), to, adjoint=True, method="rk4", options=dict(step_size=0.25))
    (ref * lw.double()).sum().backward()
    grads = []
    for need_t in (True, False):
        func = LinearField(H, C, scale=0.3, seed=2).to(DEV)
        X = native.LinearInterpolation(x.to(DEV), knots.to(DEV))
        td = t_out.to(DEV).requires_grad_(need_t)
        out = native.cdeint(X, func, z0.to(DEV), td, method="rk4", options=dict(step_size=0.25))
        (out * lw.to(DEV)).sum().backward()
        if need_t:
            _close(td.grad, to.grad, 2e-3, 2e-3 * to.grad.abs().max().item())
        grads.append(func.linear.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])

    # arbitrary func (sigmoid field of the reference's tests), float64, cubic control: step-wise continuous adjoint
    class _Sig(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.variable = torch.nn.Parameter(torch.linspace(-0.5, 0.5, 3, dtype=torch.float64).view(1, 1, 3))

        def forward(self, t, z):
            return z.sigmoid().unsqueeze(-1) + self.variable

    xs = make_series(5, 9, 3, torch.float64, seed=8)
    cs = oracle_interp.hermite_bdiff_coeffs(xs)
    zs = torch.randn(5, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(8))
    ts = torch.tensor([0., 3.3, 8.], dtype=torch.float64)
    fo, fd = _Sig(), _Sig().to(DEV)
    tso = ts.clone().requires_grad_(True)
    refs = oracle_cde.cdeint(oracle_interp.CubicPath(cs), fo, zs, tso, adjoint=True, method="rk4",
                             options=dict(step_size=0.5))
    refs.sum().backward()
    variable_grads = []
    for need_t in (True, False):
        fd.zero_grad()
        tsd = ts.to(DEV).requires_grad_(need_t)
        outs = native.cdeint(native.CubicSpline(cs.to(DEV)), fd, zs.to(DEV), tsd, method="rk4", options=dict(step_size=0.5))
        outs.sum().backward()
        if need_t:
            _close(outs, refs, 1e-9, 1e-11)
            _close(tsd.grad, tso.grad, 1e-7, 1e-9)
            _close(fd.variable.grad, fo.variable.grad, 1e-7, 1e-9)
        variable_grads.append(fd.variable.grad.clone())
    assert torch.equal(variable_grads[0], variable_grads[1])


def test_stacked_cdes_propagate_gradients_once(native):
    """reference test/test_tricks.py:54-106: the output path of one CDE (many output times) becomes the control of a
    second one; gradients must reach the first CDE's func and data, passing each intermediate exactly once."""
    class Record(torch.autograd.Function):
        @staticmethod
        def forward(ctx, name, v):
            ctx.name = name
            return v.view_as(v)

        @staticmethod
        def backward(ctx, g):
            assert not hasattr(ctx, "been_here_before"), ctx.name
            ctx.been_here_before = True
            return None, g

    fits = [(native.linear_interpolation_coeffs, native.LinearInterpolation),
            (native.hermite_cubic_coefficients_with_backward_differences, native.CubicSpline)]
    for first_fit, First in fits:
        for second_fit, Second in fits:
            first_path = torch.rand(3, 40, 2, device=DEV, requires_grad=True)
            first_coeff = first_fit(first_path)
            first_X = First(first_coeff)
            first_func = LinearField(4, 2, scale=0.5, tanh=True, seed=1).to(DEV)
            second_t = torch.linspace(0, 39, 14, device=DEV)
            second_path = native.cdeint(first_X, first_func, torch.rand(3, 4, device=DEV), second_t, method="rk4",
                                        options=dict(step_size=1.0),
                                        adjoint_params=tuple(first_func.parameters()) + (first_coeff,))
            second_path = Record.apply("second", second_path)
            second_coeff = second_fit(second_path, second_t)
            second_X = Second(second_coeff, second_t)
            second_func = LinearField(3, 4, scale=0.5, tanh=True, seed=2).to(DEV)
            third_t = torch.linspace(0, 39, 5, device=DEV)
            third = native.cdeint(second_X, second_func, torch.rand(3, 3, device=DEV), third_t, method="rk4",
                                  options=dict(step_size=1.0),
                                  adjoint_params=tuple(second_func.parameters()) + (second_coeff,))
            third = Record.apply("third", third)
            assert first_func.linear.weight.grad is None and first_path.grad is None
            third[:, -1].sum().backward()
            for g in (second_func.linear.weight.grad, first_func.linear.weight.grad, first_path.grad):
                assert isinstance(g, torch.Tensor) and torch.isfinite(g).all() and g.abs().sum() > 0


def test_control_gradients_are_not_dropped_with_a_frozen_field(native):
    """adjoint_params = (coeffs,) with a frozen func: the default dopri5 solve and the two-layer field must still
    deliver dL/dcoeffs (they route to the step-wise solver) -- compared with the fused rk4 gradient."""
    B, L, C, H = 20, 10, 4, 8
    x = make_series(B, L, C, seed=5).to(DEV)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(5)).to(DEV)
    for make in (lambda: LinearField(H, C, scale=0.4, tanh=True, seed=3), lambda: _TwoLayerField(H, C, 32, seed=3)):
        got = {}
        for method, kw in (("rk4", dict(options=dict(step_size=0.125))), ("dopri5", dict(rtol=1e-6, atol=1e-8))):
            func = make().to(DEV)
            for p in func.parameters():
                p.requires_grad_(False)
            coeffs = native.hermite_cubic_coefficients_with_backward_differences(x).requires_grad_(True)
            X = native.CubicSpline(coeffs)
            out = native.cdeint(X, func, z0, X.interval, method=method, adjoint_params=(coeffs,), **kw)
            assert out.requires_grad, method
            out[:, -1].sum().backward()
            assert coeffs.grad is not None and coeffs.grad.abs().sum() > 0, method
            got[method] = coeffs.grad.clone()
        # two different quadratures of the same continuous gradient (per-interval moments of a^T F): plausibility only
        _close(got["dopri5"], got["rk4"], 0.1, 1e-2 * got["rk4"].abs().max().item())


@pytest.mark.parametrize("H,C,width", [(32, 8, 128), (16, 14, 100)])
@pytest.mark.parametrize("degree", [3, 1])
def test_two_layer_rk4_output_time_gradients(native, degree, H, C, width):
    """Output-time gradients of the examples' two-layer model under rk4, fused (K2m + K3m): dL/dt_i = f . dL/dz_i on the
    host, dL/dt_0 from the control gradient the sweep accumulates (cubic control) or without it (piecewise-linear: no
    d2X/dt2 term) -- against the float64 oracle's odeint_adjoint (torchdiffeq's time_vjps restated).  Both tile layouts
    (32 units x 8 channels; 16 x 16 with the 14 channels of config 5's logsignature control)."""
    x, knots, coeffs, z0, t_out, lw = _time_grad_case(torch.float32, B=45, C=C, H=H)
    f64 = _TwoLayerField(H, C, width, torch.float64, seed=4)
    Xo = (oracle_interp.CubicPath(coeffs.double(), knots.double()) if degree == 3
          else oracle_interp.LinearPath(x.double(), knots.double()))
    zo = z0.double().requires_grad_(True)
    to = t_out.double().requires_grad_(True)
    ref = oracle_cde.cdeint(Xo, f64, zo, to, adjoint=True, method="rk4", options=dict(step_size=0.25))
    (ref * lw.double()).sum().backward()
    func = _TwoLayerField(H, C, width, seed=4).to(DEV)
    X = (native.CubicSpline(coeffs.to(DEV), knots.to(DEV)) if degree == 3
         else native.LinearInterpolation(x.to(DEV), knots.to(DEV)))
    zd = z0.to(DEV).requires_grad_(True)
    td = t_out.to(DEV).requires_grad_(True)
    out = native.cdeint(X, func, zd, td, method="rk4", options=dict(step_size=0.25))
    _expect_dispatch("two_layer_rk4_times", out)                          # no step-wise path
    (out * lw.to(DEV)).sum().backward()
    _close(out, ref, 1e-4, 2e-5)
    _close(zd.grad, zo.grad, 2e-3, 1e-3 * zo.grad.abs().max().item())
    _close(td.grad, to.grad, 2e-3, 2e-3 * to.grad.abs().max().item())
    for (name, got), want in zip(func.named_parameters(), f64.parameters()):
        _close(got.grad, want.grad, 2e-3, 2e-3 * want.grad.abs().max().item())
    # the reference's detach trick (test/test_tricks.py:111-131): the other gradients do not change when `t` needs none
    with_t = [p.grad.clone() for p in func.parameters()] + [zd.grad.clone()]
    func.zero_grad()
    zd.grad = None
    out = native.cdeint(X, func, zd, t_out.to(DEV), method="rk4", options=dict(step_size=0.25))
    (out * lw.to(DEV)).sum().backward()
    for a, b in zip(with_t, [p.grad for p in func.parameters()] + [zd.grad]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("H,C,width", [(32, 8, 128), (16, 14, 100)])
@pytest.mark.parametrize("degree", [3, 1])
def test_two_layer_rk4_knot_time_and_control_gradients(native, degree, H, C, width):
    """reference test/test_tricks.py:21-49 with the examples' two-layer model under rk4: adjoint_params = the model's
    parameters + the control's tensors -- knot times, and (piecewise-linear control) the knot values -- with output times
    that require a gradient as well; nothing runs step-wise.  Against the float64 oracle's odeint_adjoint.  Both tile layouts."""
    x, knots, coeffs, z0, t_out, lw = _time_grad_case(torch.float32, B=45, C=C, H=H)
    f64 = _TwoLayerField(H, C, width, torch.float64, seed=4)
    kn = knots.double().requires_grad_(True)
    xo = x.double().requires_grad_(True)
    Xo = oracle_interp.CubicPath(coeffs.double(), kn) if degree == 3 else oracle_interp.LinearPath(xo, kn)
    to = t_out.double().requires_grad_(True)
    zo = z0.double().requires_grad_(True)
    extra = (kn,) if degree == 3 else (xo, kn)
    ref = oracle_cde.cdeint(Xo, f64, zo, to, adjoint=True, method="rk4", options=dict(step_size=0.25),
                            adjoint_params=tuple(f64.parameters()) + extra)
    (ref * lw.double()).sum().backward()

    func = _TwoLayerField(H, C, width, seed=4).to(DEV)
    kd = knots.to(DEV).requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    X = native.CubicSpline(coeffs.to(DEV), kd) if degree == 3 else native.LinearInterpolation(xd, kd)
    td = t_out.to(DEV).requires_grad_(True)
    zd = z0.to(DEV).requires_grad_(True)
    out = native.cdeint(X, func, zd, td, method="rk4", options=dict(step_size=0.25),
                        adjoint_params=tuple(func.parameters()) + ((kd,) if degree == 3 else (xd, kd)))
    _expect_dispatch("two_layer_rk4_control", out)
    (out * lw.to(DEV)).sum().backward()
    _close(out, ref, 1e-4, 2e-5)
    _close(td.grad, to.grad, 2e-3, 2e-3 * to.grad.abs().max().item())
    _close(kd.grad, kn.grad, 2e-3, 2e-3 * kn.grad.abs().max().item())
    if degree == 1:
        _close(xd.grad, xo.grad, 2e-3, 2e-3 * xo.grad.abs().max().item())
    _close(zd.grad, zo.grad, 2e-3, 1e-3 * zo.grad.abs().max().item())
    for (name, got), want in zip(func.named_parameters(), f64.parameters()):
        _close(got.grad, want.grad, 2e-3, 2e-3 * want.grad.abs().max().item())
