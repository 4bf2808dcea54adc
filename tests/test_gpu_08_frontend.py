"""GPU parity tests through the C ABI -- Rows a8, b, f1 (generic path): the cdeint front-end, dispatch, the step-wise path for arbitrary funcs.

Tolerances and helpers: tests/gpu_common.py.  Collection order is the file order (01 first): the tests with the least driver history run first, so a failure elsewhere cannot hide them.
"""
import os

import pytest
import torch

from gpu_common import (_expect_dispatch, oracle_cde, oracle_interp, LinearField, _TwoLayerField, make_series, DEV, _close, _Mlp, _front)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,C,variant,act", [(32, 8, "mfma", False), (32, 8, "split", False), (32, 8, "auto", True),
                                             (64, 8, "auto", False), (32, 16, "auto", True), (20, 5, "generic", False)])
def test_solver_calls_are_graph_capturable(native, H, C, variant, act):
    """The C ABI promises that no entry point synchronises a stream or copies to the host (include/cde_mi355x.h; round 2's
    wide adjoint read `seg_off` back and synchronised -- it now takes the host copy of the offsets).  Proof by
    construction: cde_rk4_forward_linear + cde_rk4_adjoint_linear of every kernel family (MFMA, workgroup-per-tile,
    wide 8x2 / 4x4 with their host-side chunk loop, generic) are captured into a hipGraph -- a capture fails on any
    synchronising call -- and the replayed graph reproduces the eagerly computed results bit for bit, twice."""
    import importlib
    from torchcde_amd import fields
    cd = importlib.import_module("torchcde_amd.cdeint")      # the package attribute `cdeint` is the function
    B, L = 70, 9
    x = make_series(B, L, C, torch.float32, seed=300 + H).to(DEV)
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x))
    func = LinearField(H, C, torch.float32, scale=0.3, tanh=act, seed=5).to(DEV)
    gen = torch.Generator().manual_seed(H * C)
    z0 = torch.randn(B, H, generator=gen).to(DEV)
    t_out = torch.tensor([0., 3.5, 8.])
    go = (torch.rand(B, 3, H, generator=gen) + 0.5).to(DEV)
    field, _ = fields.probe(func, t_out[0].to(DEV), z0)
    code = {"auto": cd._lib.VARIANT_AUTO, "generic": cd._lib.VARIANT_GENERIC, "mfma": cd._lib.VARIANT_MFMA,
            "split": cd._lib.VARIANT_SPLIT}[variant]
    plan = cd._Plan(X, field, (B,), H, C, t_out, 1.0, 1.0, True, code)
    w, b = func.linear.weight.detach(), func.linear.bias.detach()

    def step():
        out = plan.run_forward(z0, w, b)
        gz, gw, gb, _ = plan.run_adjoint(out, go, w, b)
        return out, gz, gw.clone(), gb.clone()

    eager = step()                                             # also warms the host-side caches (grids, host copies)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                                 # warm-up on the capture stream (lazy kernel attributes)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            captured = step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        for tensor in captured:
            tensor.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for got, want in zip(captured, eager):
            assert torch.equal(got, want), (H, C, variant)


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

    class Decay:                                  # a non-Module `func.prod` with adjoint_params=() (solver.py:159-165)
        def prod(self, t, z, dXdt):
            return -z * dXdt[..., :1]

    with torch.no_grad():
        decayed = native.cdeint(X, Decay(), z0, X.interval, method="rk4", options=dict(step_size=0.5), adjoint_params=())
    assert decayed.shape == (4, 2, 3) and torch.isfinite(decayed).all()


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


def test_func_prod_interface(native):
    """`func.prod(t, z, dXdt)` (reference solver.py:48-53, :121-123): the user computes f(t, z) dX/dt itself.  Solved
    step by step; must agree with the fused solve of the same field given through `forward`, gradients included."""
    B, L, C, H = 33, 11, 4, 9

    class ProdField(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inner = LinearField(H, C, scale=0.3, tanh=True, seed=12)

        def prod(self, t, z, dXdt):
            return (self.inner(t, z) @ dXdt.unsqueeze(-1)).squeeze(-1)

    x = make_series(B, L, C, seed=111).to(DEV)
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x))
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(112)).to(DEV)
    kw = dict(method="rk4", options=dict(step_size=1.0))
    prod, plain = ProdField().to(DEV), LinearField(H, C, scale=0.3, tanh=True, seed=12).to(DEV)
    za, zb = z0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
    out_a = native.cdeint(X, prod, za, X.interval, **kw)
    out_b = native.cdeint(X, plain, zb, X.interval, **kw)
    _close(out_a, out_b, 1e-5, 1e-6)
    out_a[:, -1].square().sum().backward()
    out_b[:, -1].square().sum().backward()
    _close(za.grad, zb.grad, 1e-4, 1e-5)
    _close(prod.inner.linear.weight.grad, plain.linear.weight.grad, 1e-4, 1e-4 * plain.linear.weight.grad.abs().max().item())

    class BadProd(torch.nn.Module):
        def prod(self, t, z, dXdt):
            return z[..., :-1]

    with pytest.raises(ValueError, match="func.prod did not return a tensor with the same shape as z0"):
        native.cdeint(X, BadProd(), z0, X.interval, **kw)


def test_tuple_state_and_tuple_control(native):
    """Tuple-valued state with TupleControl (reference solver.py:68-95, misc.py:129-164): solved on the concatenated
    state by the step-wise path; with an uncoupled func it must reproduce the two separate (fused) solves, gradients
    included.  Error behaviour as the reference."""
    B, L = 21, 9
    x1, x2 = make_series(B, L, 3, seed=131).to(DEV), make_series(B, L, 5, seed=132).to(DEV)
    X1 = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x1))
    X2 = native.LinearInterpolation(native.linear_interpolation_coeffs(x2))
    X = native.TupleControl(X1, X2)
    assert torch.equal(X.interval, X1.interval) and torch.equal(X.grid_points, X1.grid_points)
    dX = X.derivative(torch.tensor(2.5, device=DEV))
    assert isinstance(dX, tuple) and dX[0].shape == (B, 3) and dX[1].shape == (B, 5)

    class Pair(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.f1 = LinearField(6, 3, scale=0.3, tanh=True, seed=13)
            self.f2 = LinearField(4, 5, scale=0.3, seed=14)

        def forward(self, t, z):
            return self.f1(t, z[0]), self.f2(t, z[1])

    gen = torch.Generator().manual_seed(133)
    z1, z2 = torch.randn(B, 6, generator=gen).to(DEV), torch.randn(B, 4, generator=gen).to(DEV)
    t = torch.tensor([0., 3., 8.], device=DEV)
    kw = dict(method="rk4", options=dict(step_size=1.0))
    pair = Pair().to(DEV)
    a1, a2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    out = native.cdeint(X, pair, (a1, a2), t, **kw)
    assert isinstance(out, tuple) and out[0].shape == (B, 3, 6) and out[1].shape == (B, 3, 4)
    (out[0].square().sum() + out[1].square().sum()).backward()

    f1, f2 = LinearField(6, 3, scale=0.3, tanh=True, seed=13).to(DEV), LinearField(4, 5, scale=0.3, seed=14).to(DEV)
    b1, b2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    o1 = native.cdeint(X1, f1, b1, t, **kw)
    o2 = native.cdeint(X2, f2, b2, t, **kw)
    (o1.square().sum() + o2.square().sum()).backward()
    _close(out[0], o1, 1e-5, 1e-6)
    _close(out[1], o2, 1e-5, 1e-6)
    _close(a1.grad, b1.grad, 1e-4, 1e-5)
    _close(a2.grad, b2.grad, 1e-4, 1e-5)
    _close(pair.f1.linear.weight.grad, f1.linear.weight.grad, 1e-4, 1e-4 * f1.linear.weight.grad.abs().max().item())
    _close(pair.f2.linear.bias.grad, f2.linear.bias.grad, 1e-4, 1e-4 * f2.linear.bias.grad.abs().max().item())

    with pytest.raises(ValueError, match="must be tuples of the same length"):
        native.cdeint(X, pair, (z1,), t, **kw)
    with pytest.raises(ValueError, match="X.derivative must return a tuple/list"):
        native.cdeint(X1, pair, (z1, z2), t, **kw)
    with pytest.raises(ValueError, match="one or more controls"):
        native.TupleControl()
    short = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x1[:, :5]))
    with pytest.raises(ValueError, match="same interval"):
        native.TupleControl(X1, short)


@pytest.mark.parametrize("kind", ["two_layer", "tanh_wide"])
def test_stepwise_adjoint_in_closed_form_for_recognised_fields(native, kind):
    """The reference's default call (dopri5 + adjoint) where no fused backward kernel applies -- two-layer fields, affine
    fields beyond the 32 x 8 tiles: the step-wise adjoint writes the augmented dynamics of a RECOGNISED field in closed
    form (stepwise._explicit_dynamics) instead of differentiating func with autograd.  Against the autograd route
    (`variant="generic"`) and the float64 oracle; output-time gradients included."""
    from torchcde_amd import stepwise
    B, L = 7, 5
    if kind.startswith("two_layer"):
        H, C = 8, 3
        make = lambda dtype: _TwoLayerField(H, C, 24, dtype, seed=5, final_tanh=kind == "two_layer")
    else:
        H, C = 40, 5
        make = lambda dtype: LinearField(H, C, dtype, scale=0.3, tanh=kind == "tanh_wide", seed=5)
    x = make_series(B, L, C, torch.float32, seed=31)
    coeffs = oracle_interp.hermite_bdiff_coeffs(x)
    gen = torch.Generator().manual_seed(32)
    z0 = torch.randn(B, H, generator=gen)
    t_out = torch.tensor([0., 1.4, 4.])
    lw = torch.rand(B, 3, H, generator=gen) + 0.5
    f64 = make(torch.float64)
    zo = z0.double().requires_grad_(True)
    to = t_out.double().requires_grad_(True)
    # a request that stays step-wise for EVERY recognised field: max_num_steps is a torchdiffeq option no fused kernel
    # takes (dispatch row "adjoint_options outside the fused kernels' set"); it changes no arithmetic.  The plain call of
    # the two-layer field is fused since round 4 (`mlp_dopri5_adjoint`, DISPATCH_CASES["two_layer_default_call_wants_t"]).
    extra = dict(adjoint_options=dict(max_num_steps=10 ** 6))
    ref = oracle_cde.cdeint(oracle_interp.CubicPath(coeffs.double()), f64, zo, to, adjoint=True, rtol=1e-5, atol=1e-7, **extra)
    (ref * lw.double()).sum().backward()
    calls = []
    original = stepwise._explicit_dynamics

    def spy(field, params):
        run = original(field, params)
        calls.append(run is not None)
        return run

    stepwise._explicit_dynamics = spy
    try:
        results = {}
        for variant, need_t in (("auto", True), ("generic", True)):
            func = make(torch.float32).to(DEV)
            X = native.CubicSpline(coeffs.to(DEV))
            zd = z0.to(DEV).requires_grad_(True)
            td = t_out.to(DEV).requires_grad_(need_t)
            out = native.cdeint(X, func, zd, td, rtol=1e-5, atol=1e-7, variant=variant, **extra)
            _expect_dispatch("two_layer_adjoint_options_beyond_the_kernels" if kind.startswith("two_layer")
                             else "adjoint_options_beyond_the_kernels")
            (out * lw.to(DEV)).sum().backward()
            results[(variant, need_t)] = (out.detach(), zd.grad, td.grad, [p.grad.clone() for p in func.parameters()])
    finally:
        stepwise._explicit_dynamics = original
    assert calls == [True, False]                             # closed form under AUTO, autograd under "generic"
    out, gz, gt, gp = results[("auto", True)]
    # closed form vs autograd: the same float32 solver on the same dynamics (round-off apart)
    ga = results[("generic", True)]
    _close(out, ga[0], 2e-3, 2e-3 * ga[0].abs().max().item())
    _close(gz, ga[1], 5e-3, 5e-3 * ga[1].abs().max().item())
    _close(gt, ga[2], 5e-3, 5e-3 * ga[2].abs().max().item())
    for a, b in zip(gp, ga[3]):
        _close(a, b, 5e-3, 5e-3 * max(1e-3, b.abs().max().item()))
    # vs the float64 oracle: two adaptive solves at rtol 1e-5 agree to a few 1e-3 of the largest entry
    _close(out, ref, 2e-2, 2e-2 * ref.abs().max().item())
    _close(gz, zo.grad, 2e-2, 2e-2 * zo.grad.abs().max().item())
    _close(gt, to.grad, 2e-2, 2e-2 * to.grad.abs().max().item())
    for got, p64 in zip(gp, f64.parameters()):
        _close(got, p64.grad, 2e-2, 2e-2 * max(1e-3, p64.grad.abs().max().item()))


def test_wide_affine_fields_beyond_the_fused_tiles(native):
    """H = 32 with C = 64 (and logsignature-sized C = 204): the generic kernel's LDS tile no longer holds the default
    number of series -- the solve must still run forward AND backward (smaller tiles or the step-wise path), not fail
    in backward()."""
    for C, B in ((64, 6), (204, 3)):
        H, L = 32, 6
        x = make_series(B, L, C, seed=C)
        func = LinearField(H, C, scale=0.05, seed=1)
        z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(1))
        f64 = LinearField(H, C, torch.float64, scale=0.05, seed=1)
        zo = z0.double().requires_grad_(True)
        Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x.double()))
        ref = oracle_cde.cdeint(Xo, f64, zo, Xo.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
        ref[:, -1].sum().backward()
        fd = func.to(DEV)
        zd = z0.to(DEV).requires_grad_(True)
        X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV)))
        out = native.cdeint(X, fd, zd, X.interval, method="rk4", options=dict(step_size=1.0))
        _close(out, ref, 1e-4, 1e-5)
        out[:, -1].sum().backward()
        _close(zd.grad, zo.grad, 1e-3, 1e-5)
        _close(fd.linear.weight.grad, f64.linear.weight.grad, 1e-3, 1e-3 * f64.linear.weight.grad.abs().max().item())


def test_dispatch_is_queryable_and_warns_once_when_a_known_field_goes_stepwise(native):
    """no silent 1000x cliffs.  Every cdeint call leaves (path, reason) in
    `last_dispatch()`; a RECOGNISED field that leaves the fused kernels warns once per (module class, reason); and options
    the step-wise solver supports but the fused kernels do not (`max_num_steps`) are routed there instead of raising
    (the reference forwards all options verbatim, solver.py:175-176,227)."""
    import warnings
    front = _front()
    B, L, C, H = 40, 8, 4, 12
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(make_series(B, L, C, seed=3).to(DEV)))

    class Field(LinearField):            # a class of its own: the once-per-class registry is global
        pass
    func = Field(H, C, scale=0.3, seed=1).to(DEV)
    z0 = torch.randn(B, H, device=DEV)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            fused = native.cdeint(X, func, z0, X.interval, method="rk4", options=dict(step_size=1.0))
        assert front.last_dispatch()[0] == ("rk4", "")
        with torch.no_grad():
            native.cdeint(X, func, z0, X.interval)
        assert front.last_dispatch()[0].path == "dopri5_forward"
        assert not [w for w in caught if "step-wise" in str(w.message)]
        with torch.no_grad():
            capped = native.cdeint(X, func, z0, X.interval, options=dict(max_num_steps=10000), rtol=1e-6, atol=1e-8)  # round 2: raised
        choice, request = front.last_dispatch()
        assert choice.path == "stepwise" and "options" in choice.reason and request.kind == "affine"
        with torch.no_grad():
            native.cdeint(X, func, z0, X.interval, options=dict(max_num_steps=10000))
            native.cdeint(X, func, z0, X.interval, method="midpoint", options=dict(step_size=0.5))
        assert front.last_dispatch()[0] == ("fixed_grid", "")                  # midpoint / euler are fused since round 5 ...
        # ... for adjoint=True: backpropagating through a midpoint solve is still the step-wise path, and says so
        native.cdeint(X, func, z0.clone().requires_grad_(True), X.interval, method="midpoint", adjoint=False,
                      options=dict(step_size=0.5))
        assert "midpoint" in front.last_dispatch()[0].reason
    told = [str(w.message) for w in caught if "step-wise" in str(w.message)]
    assert len(told) == 2 and "Field" in told[0] and "options" in told[0] and "midpoint" in told[1]      # once per reason
    with torch.no_grad():
        plain = native.cdeint(X, func, z0, X.interval, rtol=1e-6, atol=1e-8)
    _close(capped, plain, 1e-3, 1e-4)                          # the same adaptive solve, host-driven vs fused
    assert torch.isfinite(fused).all()
    with pytest.raises(AssertionError, match="max_num_steps"):
        with torch.no_grad():
            native.cdeint(X, func, z0, X.interval, options=dict(max_num_steps=2))


class _SineControl(torch.nn.Module):
    """A user-defined control: X_c(t) = amp_c sin(w_c t + phase_c) per series -- nothing but a `derivative` method, which is all
    reference solver.py:45-46 asks of X (no coefficients, no grid points, not one of the package's path classes)."""

    def __init__(self, B, C, dtype, seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.amp = torch.nn.Parameter(torch.rand(B, C, generator=gen, dtype=dtype) + 0.5)
        self.register_buffer("freq", torch.rand(B, C, generator=gen, dtype=dtype) * 2 + 0.5)
        self.register_buffer("phase", torch.rand(B, C, generator=gen, dtype=dtype) * 6)

    def derivative(self, t):
        return self.amp * self.freq * torch.cos(self.freq * t + self.phase)


@pytest.mark.parametrize("method,adjoint", [("rk4", True), ("rk4", False), ("dopri5", True)])
def test_user_defined_control_with_only_a_derivative_method(native, method, adjoint):
    """reference solver.py:45-46: cdeint needs nothing of X but `derivative`.  A control module of the user's own runs
    step-wise on the GPU (its derivative called at every evaluation, the product in cde_contract) and matches the oracle's
    cdeint over the same control: trajectories, dL/dz0, the field's parameters and the CONTROL's own parameter (passed in
    adjoint_params, README.md:251-270's pattern; under adjoint=False autograd reaches it by itself)."""
    B, C, H = 5, 3, 6
    dtype = torch.float64
    t = torch.tensor([0.0, 0.7, 1.9], dtype=dtype)
    z0 = torch.randn(B, H, dtype=dtype, generator=torch.Generator().manual_seed(5))
    kw = dict(method=method, adjoint=adjoint)
    if method == "rk4":
        kw["options"] = dict(step_size=0.1)
    else:
        kw.update(rtol=1e-8, atol=1e-10)
    res = {}
    for where in ("oracle", "native"):
        dev = "cpu" if where == "oracle" else DEV
        X = _SineControl(B, C, dtype, seed=2).to(dev)
        f = _Mlp(C, H, 16, dtype, seed=4).to(dev)
        z = z0.detach().clone().to(dev).requires_grad_(True)
        extra = dict(adjoint_params=tuple(f.parameters()) + (X.amp,)) if adjoint else {}
        solve = oracle_cde.cdeint if where == "oracle" else native.cdeint
        out = solve(X, f, z, t.to(dev), **kw, **extra)
        assert out.shape == (B, 3, H)
        out[:, 1:].pow(2).sum().backward()
        res[where] = [out.detach().cpu(), z.grad.cpu(), X.amp.grad.cpu()] + [p.grad.cpu() for p in f.parameters()]
    assert _front().last_dispatch()[0].path == "stepwise"
    tight = method == "rk4"
    for got, want in zip(res["native"], res["oracle"]):
        # (dopri5: two adaptive solves whose libm differs in the last bit take slightly different steps -- they agree at the
        #  level of the tolerance they were asked for, observed 1.3e-6 of the largest entry)
        _close(got, want, 1e-9 if tight else 2e-5, (1e-11 if tight else 2e-5) * max(1.0, want.abs().max().item()))
    # the reference's own complaints about a control that does not fit (solver.py:7-33, :56-57)
    class NoDerivative(torch.nn.Module):
        pass
    with pytest.raises(ValueError, match="X must have a 'derivative' method"):
        native.cdeint(NoDerivative(), _Mlp(C, H, 16, dtype, seed=4).to(DEV), z0.to(DEV), t.to(DEV))
    with pytest.raises(ValueError, match="same number of batch dimensions as z0"):
        native.cdeint(_SineControl(B + 1, C, dtype).to(DEV), _Mlp(C, H, 16, dtype, seed=4).to(DEV), z0.to(DEV), t.to(DEV))
    with pytest.raises(ValueError, match="same number of input channels as X.derivative"):
        native.cdeint(_SineControl(B, C + 1, dtype).to(DEV), _Mlp(C, H, 16, dtype, seed=4).to(DEV), z0.to(DEV), t.to(DEV))


class _ExampleNeuralCDE(torch.nn.Module):
    """The model of reference example/time_series_classification.py:54-94 restated over any implementation of the torchcde
    API (`api` = torchcde_amd on the GPU, the oracle's classes on the CPU): initial = Linear(C, H) on X.evaluate(X.interval[0]),
    z_T = cdeint(X, func, z0, X.interval) with NO method (dopri5 + adjoint, solver.py:195-203,226), readout = Linear(H, 1)."""

    def __init__(self, api, C, H, dtype, seed, **solver):
        super().__init__()
        self.api, self.solver = api, solver
        from helpers import TwoLayerField
        self.func = TwoLayerField(H, C, 128, dtype, seed=seed)
        gen = torch.Generator().manual_seed(seed + 1)
        self.initial = torch.nn.Linear(C, H).to(dtype)
        self.readout = torch.nn.Linear(H, 1).to(dtype)
        with torch.no_grad():
            for lin in (self.initial, self.readout):
                bound = 1 / lin.in_features ** 0.5
                lin.weight.copy_((torch.rand(lin.weight.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
                lin.bias.copy_((torch.rand(lin.bias.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)

    def forward(self, coeffs):
        X = self.api["spline"](coeffs)
        z0 = self.initial(X.evaluate(X.interval[0]))
        z_T = self.api["cdeint"](X=X, z0=z0, func=self.func, t=X.interval, **self.solver)
        return self.readout(z_T[:, 1]).squeeze(-1)


@pytest.mark.parametrize("solver,bar", [({}, 3e-2), (dict(rtol=1e-5, atol=1e-7), 2e-3)])
def test_example_neural_cde_trains_like_the_reference_model(native, solver, bar):
    """SURVEY section 2 row 12 / example/time_series_classification.py:54-147: the example's NeuralCDE on top of this package
    -- spirals (t, x, y) -> hermite coefficients -> CubicSpline -> X.evaluate(X.interval[0]) -> initial -> DEFAULT cdeint
    (dopri5 + adjoint, fused: K4 + K4am) -> readout -> BCE-with-logits -> Adam.step(), three steps at the example's batch size
    (32) -- against the same model over the oracle in float64.  Both take tolerance-level adaptive solutions with their own
    step sequences: as the example calls it (rtol 1e-4, atol 1e-6: the defaults of solver.py:195-198) the losses and predictions
    are compared at 3e-2 (observed: losses 1.5e-3, logits 1.5e-2 after three steps), the first gradients at 30 %; with the
    tolerances tightened to 1e-5 / 1e-7 the same quantities at 2e-3 / 2 %."""
    B, L, C, H = 32, 40, 3, 8
    gen = torch.Generator().manual_seed(11)
    t = torch.linspace(0.0, 4 * 3.141592653589793, L)
    start = torch.rand(B, generator=gen) * 2 * 3.141592653589793
    sign = torch.where(torch.arange(B) % 2 == 0, 1.0, -1.0)
    ang = start[:, None] + sign[:, None] * t[None]
    x = torch.stack([t.expand(B, L), torch.cos(ang) / (1 + 0.5 * t), torch.sin(ang) / (1 + 0.5 * t)], dim=2)
    x[..., 1:] += 0.01 * torch.randn(B, L, 2, generator=gen)
    y = (sign > 0).float()

    apis = {"oracle": dict(spline=oracle_interp.CubicPath, cdeint=oracle_cde.cdeint, fit=oracle_interp.hermite_bdiff_coeffs),
            "native": dict(spline=native.CubicSpline, cdeint=native.cdeint,
                           fit=native.hermite_cubic_coefficients_with_backward_differences)}
    log = {}
    for where, api in apis.items():
        dev, dtype = ("cpu", torch.float64) if where == "oracle" else (DEV, torch.float32)
        model = _ExampleNeuralCDE(api, C, H, dtype, seed=21, **solver).to(dev)
        coeffs = api["fit"](x.to(dev, dtype))
        target = y.to(dev, dtype)
        opt = torch.optim.Adam(model.parameters())
        losses, first_grads = [], None
        for step in range(3):
            pred = model(coeffs)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
            opt.zero_grad()
            loss.backward()
            if where == "native" and step == 0:
                _expect_dispatch("two_layer_dopri5", None)
            if first_grads is None:
                first_grads = [p.grad.detach().double().cpu().clone() for p in model.parameters()]
            opt.step()
            losses.append(loss.item())
        with torch.no_grad():
            final = model(coeffs).double().cpu()
        log[where] = (losses, first_grads, final)
    (lo, go, fo), (ln, gn, fn) = log["oracle"], log["native"]
    assert ln[2] < ln[0]                                  # it trains
    for a, b in zip(ln, lo):
        assert abs(a - b) <= bar * abs(b), (ln, lo)
    _close(fn, fo, bar, bar)
    for a, b in zip(gn, go):
        _close(a, b, 10 * bar, 10 * bar * b.abs().max().item())


@pytest.mark.gpu
def test_higher_order_gradients_fail_loudly_on_the_fused_path_and_work_step_wise(native):
    """the fused backward functions compute from detached tensors in raw kernels -- a double backward through them
    must raise (torch's once_differentiable), not return gradients with the higher-order terms silently missing.  With
    adjoint=False the reference (autograd through torchdiffeq.odeint: solver.py:144,226-227) supports create_graph=True; here
    variant="generic" keeps that request on the step-wise path, where the Hessian-vector product matches the float64 oracle's."""
    import warnings
    B, L, C, H = 12, 7, 3, 8
    x = make_series(B, L, C, seed=41)
    z0 = torch.randn(B, H, generator=torch.Generator().manual_seed(41))
    kw = dict(method="rk4", options=dict(step_size=0.5), adjoint=False)

    def hvp(cdeint, X, func, z):
        out = cdeint(X, func, z, X.interval, **kw)
        (g,) = torch.autograd.grad(out[:, -1].square().sum(), z, create_graph=True)
        return g, g.square().sum()

    func = LinearField(H, C, scale=0.4, tanh=True, seed=5).to(DEV)
    X = native.CubicSpline(native.hermite_cubic_coefficients_with_backward_differences(x.to(DEV)))
    zd = z0.to(DEV).requires_grad_(True)
    g, s = hvp(native.cdeint, X, func, zd)
    _expect_dispatch("affine_rk4_backprop")
    with pytest.raises(RuntimeError, match="once_differentiable|differentiate"):
        s.backward()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                        # (the step-wise warning)
        g2, s2 = hvp(lambda *a, **k: native.cdeint(*a, variant="generic", **k), X, func, zd)
    assert front_path() == "stepwise"
    zd.grad = None
    s2.backward()
    f64 = LinearField(H, C, torch.float64, scale=0.4, tanh=True, seed=5)
    Xo = oracle_interp.CubicPath(oracle_interp.hermite_bdiff_coeffs(x.double()))
    zo = z0.double().requires_grad_(True)
    g3, s3 = hvp(oracle_cde.cdeint, Xo, f64, zo)
    s3.backward()
    _close(g2, g3, 1e-3, 1e-4 * g3.abs().max().item())
    _close(zd.grad, zo.grad, 2e-3, 2e-4 * zo.grad.abs().max().item())


def front_path():
    return _front().last_dispatch()[0].path
