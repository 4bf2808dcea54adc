"""Generate tests/golden/*.pt from the REAL reference (run in the build container only).

    python oracle/make_golden.py                        # (re)write the fixtures + bitwise self-check
    python oracle/make_golden.py --run-reference-tests  # additionally run the reference's own pytest files

/root/reference is imported read-only.  Its ``torchcde/__init__.py:7`` imports
``solver.py``, which imports ``torchdiffeq`` and ``torchsde`` (solver.py:2-3); neither is
installed, so before importing we register ``oracle.odeint`` as ``torchdiffeq`` and an
empty module as ``torchsde``.  Consequences, stated once:

  * interpolation fixtures (hermite / spline / linear) are produced by reference code only
    -> they PIN ``oracle.interp`` (this script also asserts bitwise equality oracle vs reference);
  * ``cdeint`` fixtures are produced by the reference's real ``solver.py`` (vector field,
    defaults, permute) driving ``oracle.odeint`` -> they pin ``oracle.cde`` and the plumbing, but
    the integrator arithmetic itself stays PARITY UNPINNED (see oracle/odeint.py).

Nothing here runs on the GPU box (no /root/reference there); tests read only the .pt files.
"""
import argparse
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    from oracle import odeint as oracle_ode
    shim = types.ModuleType("torchdiffeq")
    shim.odeint = oracle_ode.odeint
    shim.odeint_adjoint = oracle_ode.odeint_adjoint
    sys.modules["torchdiffeq"] = shim
    sys.modules["torchsde"] = types.ModuleType("torchsde")
    from oracle import logsig as oracle_logsig
    sys.modules["signatory"] = oracle_logsig.as_signatory_module()      # log_ode.py:1-8 imports it at module load
    sys.path.insert(0, REFERENCE)
    import torchcde
    assert os.path.realpath(torchcde.__file__).startswith(REFERENCE)
    return torchcde


def _irregular_t(L, dtype, gen):
    gaps = torch.rand(L, generator=gen, dtype=torch.float64) * 1.5 + 0.1
    return gaps.cumsum(0).to(dtype)


def _query_times(knots, gen):
    lo, hi = knots[0].item(), knots[-1].item()
    inside = torch.rand(9, generator=gen, dtype=torch.float64) * (hi - lo) + lo
    extra = torch.tensor([lo - 0.7, lo, hi, hi + 1.3], dtype=torch.float64)
    on_knots = knots[: min(4, len(knots))].to(torch.float64)
    return torch.cat([inside, extra, on_knots]).to(knots.dtype)


def interpolation_cases(ref):
    from oracle import interp
    gen = torch.Generator().manual_seed(20240925)
    cases = []
    shapes = [((1,), 2, 1), ((3,), 5, 3), ((2, 3), 10, 6), ((4,), 16, 8), ((), 7, 2), ((5,), 128, 8)]
    for dtype in (torch.float32, torch.float64):
        for batch, L, C in shapes:
            for explicit_t in (False, True):
                x = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
                t = _irregular_t(L, dtype, gen) if explicit_t else None
                coeffs = ref.hermite_cubic_coefficients_with_backward_differences(x, t)
                assert torch.equal(coeffs, interp.hermite_bdiff_coeffs(x, t)), "oracle hermite != reference"
                spline = ref.CubicSpline(coeffs, t)
                knots = spline.grid_points
                tq = _query_times(knots, gen)
                frac, index = spline._interpret_t(tq)
                value = spline.evaluate(tq)
                slope = spline.derivative(tq)
                ofrac, oindex = interp.locate(tq, knots, coeffs.size(-2), dtype, "cpu")
                assert torch.equal(index, oindex) and torch.equal(frac, ofrac), "oracle locate != reference"
                assert torch.equal(value, interp.cubic_value(coeffs, knots, tq)), "oracle evaluate != reference"
                assert torch.equal(slope, interp.cubic_slope(coeffs, knots, tq)), "oracle derivative != reference"
                # scalar-time call path (what the solver uses)
                s_val = torch.stack([spline.evaluate(q) for q in tq], dim=-2)
                s_der = torch.stack([spline.derivative(q) for q in tq], dim=-2)
                assert torch.equal(s_val, value) and torch.equal(s_der, slope)
                # piecewise-linear control on the same data
                lin_coeffs = ref.linear_interpolation_coeffs(x, t)
                assert lin_coeffs is x
                lin = ref.LinearInterpolation(lin_coeffs, t)
                lfrac, lindex = lin._interpret_t(tq)
                lvalue = lin.evaluate(tq)
                lslope = lin.derivative(tq)
                opath = interp.LinearPath(lin_coeffs, t)
                assert torch.equal(lvalue, opath.evaluate(tq)) and torch.equal(lslope, opath.derivative(tq))
                cases.append(dict(x=x, t=t, knots=knots.clone(), coeffs=coeffs, tq=tq, index=index, frac=frac,
                                  value=value, slope=slope, lin_index=lindex, lin_frac=lfrac, lin_value=lvalue,
                                  lin_slope=lslope))
    return cases


def nan_fill_cases(ref):
    """Missing-value construction (reference interpolation_linear.py:13-84) incl. the edge cases its own test
    exercises (test/test_linear_interpolation.py:6-48): leading / trailing gaps, long runs, all-NaN channels."""
    from oracle import interp
    gen = torch.Generator().manual_seed(4242)
    cases = []
    for dtype in (torch.float32, torch.float64):
        for batch, L, C, p_nan, explicit_t in (((3,), 7, 2, 0.3, False), ((2, 2), 12, 3, 0.5, True),
                                               ((4,), 20, 8, 0.15, True), ((1,), 5, 1, 0.6, False)):
            x = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
            mask = torch.rand(*batch, L, C, generator=gen) < p_nan
            x = x.masked_fill(mask, float("nan"))
            flat = x.view(-1, L, C)
            flat[0, :, 0] = float("nan")                       # an all-NaN scalar path
            flat[-1, :2, -1] = float("nan")                    # leading gap
            flat[-1, -2:, -1] = float("nan")                   # trailing gap
            t = _irregular_t(L, dtype, gen) if explicit_t else None
            filled = ref.linear_interpolation_coeffs(x, t)
            assert not torch.isnan(filled).any()
            assert torch.equal(filled, interp.linear_coeffs(x, t)), "oracle NaN fill != reference"
            hermite = ref.hermite_cubic_coefficients_with_backward_differences(x, t)
            assert torch.equal(hermite, interp.hermite_bdiff_coeffs(x, t))
            cases.append(dict(x=x, t=t, filled=filled, hermite=hermite))
    return cases


def rectilinear_cases(ref):
    """forward_fill (reference misc.py:103-126) and rectilinear preparation (interpolation_linear.py:86-128, :152-162),
    incl. the literal known-answer case of the reference's own test (test/test_linear_interpolation.py:117-152)."""
    from oracle import interp
    nan = float("nan")
    gen = torch.Generator().manual_seed(777)
    cases = []
    # the reference test's hand-written example: two series of lengths 3 and 2, padded with NaN, time in channel 0
    t1 = torch.tensor([0.1, 0.2, 0.9]).view(-1, 1)
    t2 = torch.tensor([0.2, 0.3]).view(-1, 1)
    x1 = torch.tensor([0.4, nan, 1.1]).view(-1, 1)
    x2 = torch.tensor([nan, 2.]).view(-1, 1)
    x = torch.nn.utils.rnn.pad_sequence([torch.cat((t1, x1), -1), torch.cat((t2, x2), -1)], batch_first=True,
                                        padding_value=nan)
    x[:, :, 0] = ref.misc.forward_fill(x[:, :, 0], fill_index=-1)
    x1_true = torch.tensor([[0.1, 0.2, 0.2, 0.9, 0.9], [0.4, 0.4, 0.4, 0.4, 1.1]]).T.view(-1, 2)
    x2_true = torch.tensor([[0.2, 0.3, 0.3, 0.3, 0.3], [2., 2., 2., 2., 2.]]).T.view(-1, 2)
    known = torch.stack((x1_true, x2_true))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ref.linear_interpolation_coeffs(x, rectilinear=0)
        assert torch.equal(known, got)
        assert torch.equal(got, interp.linear_coeffs(x, rectilinear=0))
        cases.append(dict(x=x, time_index=0, filled=ref.misc.forward_fill(x), prepared=ref.interpolation_linear
                          ._prepare_rectilinear_interpolation(x, 0), coeffs=got, known_answer=True))
        for dtype in (torch.float32, torch.float64):
            for batch, L, C, time_index, p_nan in (((5,), 9, 4, 0, 0.4), ((2, 3), 17, 6, 3, 0.25), ((1,), 2, 2, 1, 0.5),
                                                   ((7,), 33, 8, 7, 0.1)):
                data = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
                data = data.masked_fill(torch.rand(*batch, L, C, generator=gen) < p_nan, nan)
                data[..., time_index] = torch.rand(*batch, L, generator=gen, dtype=dtype).cumsum(-1)   # time: no NaN
                filled = ref.misc.forward_fill(data)
                prepared = ref.interpolation_linear._prepare_rectilinear_interpolation(data, time_index)
                coeffs = ref.linear_interpolation_coeffs(data, rectilinear=time_index)
                same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=1e30), torch.nan_to_num(b, nan=1e30)) and \
                    torch.equal(torch.isnan(a), torch.isnan(b))
                assert same(filled, interp.forward_fill(data)), "oracle forward_fill != reference"
                assert same(prepared, interp.rectilinear_prepare(data, time_index)), "oracle rectilinear != reference"
                assert torch.equal(coeffs, interp.linear_coeffs(data, rectilinear=time_index))
                cases.append(dict(x=data, time_index=time_index, filled=filled, prepared=prepared, coeffs=coeffs,
                                  known_answer=False))
    return cases


def natural_cases(ref):
    """natural_cubic_coeffs / natural_cubic_spline_coeffs (reference interpolation_cubic.py:7-266), with and without
    missing values, default and irregular t, float32 and float64, L = 2 and 3 included."""
    from oracle import interp
    gen = torch.Generator().manual_seed(991)
    cases = []
    for dtype in (torch.float32, torch.float64):
        for batch, L, C, p_nan, explicit_t in (((3,), 2, 2, 0.0, False), ((1,), 3, 1, 0.0, False), ((4,), 7, 3, 0.0, True),
                                               ((2, 2), 12, 3, 0.3, True), ((5,), 9, 2, 0.5, False),
                                               ((6,), 20, 4, 0.6, True), ((8,), 33, 8, 0.0, False)):
            x = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
            if p_nan > 0:
                x = x.masked_fill(torch.rand(*batch, L, C, generator=gen) < p_nan, float("nan"))
                x.view(-1, L, C)[0, :, 0] = float("nan")                 # an all-NaN scalar path
            t = _irregular_t(L, dtype, gen) if explicit_t else None
            v1 = ref.natural_cubic_coeffs(x, t)
            v0 = ref.natural_cubic_spline_coeffs(x, t)
            assert torch.equal(v1, interp.natural_cubic_coeffs(x, t, 1)), "oracle natural cubic (v1) != reference"
            assert torch.equal(v0, interp.natural_cubic_coeffs(x, t, 0)), "oracle natural cubic (v0) != reference"
            cases.append(dict(x=x, t=t, coeffs=v1, coeffs_v0=v0))
    return cases


def logsig_cases(ref):
    """logsig_windows / logsignature_windows (reference log_ode.py:15-133) run ON TOP OF oracle.logsig standing in for
    signatory: pins the windowing, merging, filling and accumulation of the oracle restatement; the logsignature
    arithmetic itself is parity-unpinned (see oracle/logsig.py)."""
    from oracle import logsig
    gen = torch.Generator().manual_seed(4711)
    cases = []
    for dtype in (torch.float32, torch.float64):
        for batch, L, C, depth, window, p_nan, explicit_t in (((3,), 17, 3, 3, 4.0, 0.0, False), ((2, 2), 13, 2, 2, 2.5, 0.0, False),
                                                              ((4,), 21, 3, 3, 3.0, 0.2, True), ((1,), 9, 1, 3, 8.0, 0.0, False),
                                                              ((5,), 12, 4, 1, 1.0, 0.0, False), ((2,), 30, 3, 2, 7.0, 0.3, True)):
            x = torch.randn(*batch, L, C, generator=gen, dtype=dtype).cumsum(-2) * 0.3
            if p_nan > 0:
                x = x.masked_fill(torch.rand(*batch, L, C, generator=gen) < p_nan, float("nan"))
            t = _irregular_t(L, dtype, gen) if explicit_t else None
            v1 = ref.logsig_windows(x, depth, window, t)
            v0, times0 = ref.logsignature_windows(x, depth, window, t)
            o1 = logsig.logsig_windows(x, depth, window, t, version=1)
            o0, ot = logsig.logsig_windows(x, depth, window, t, version=0)
            assert torch.equal(v1, o1) and torch.equal(v0, o0) and torch.equal(times0, ot), "oracle windowing != reference"
            cases.append(dict(x=x, t=t, depth=depth, window_length=window, out=v1, out_v0=v0, times_v0=times0))
    return cases


def logsig_rebuilt_t_cases(ref):
    """The scenario of the round-2 plan-cache bug: ONE series shape, MANY explicit time grids of the same length and dtype
    with different values (a `t` rebuilt per batch), each through the reference's log_ode.py:15-133.  A plan cached for
    one grid and served for another gives wrong windows; every output here belongs to its own grid."""
    from oracle import logsig
    gen = torch.Generator().manual_seed(20260925)
    cases = []
    for dtype in (torch.float32, torch.float64):
        L, C, depth = 33, 3, 3
        x = torch.randn(3, L, C, generator=gen, dtype=dtype).cumsum(-2) * 0.3
        grids = [torch.linspace(0, 16.0 * k, L, dtype=dtype) for k in (1, 2, 3, 5)]
        grids += [_irregular_t(L, dtype, gen) for _ in range(4)]
        for t in grids:
            for window in (4.0, 6.5):
                v1 = ref.logsig_windows(x, depth, window, t)
                v0, times0 = ref.logsignature_windows(x, depth, window, t)
                o1 = logsig.logsig_windows(x, depth, window, t, version=1)
                assert torch.equal(v1, o1), "oracle windowing != reference"
                cases.append(dict(x=x, t=t, depth=depth, window_length=window, out=v1, out_v0=v0, times_v0=times0))
    return cases


class LinearField(torch.nn.Module):
    """The README vector field (reference README.md:42-49): Linear(H, H*C) viewed (..., H, C)."""

    def __init__(self, H, C, dtype, scale, gen):
        super().__init__()
        self.H, self.C = H, C
        self.linear = torch.nn.Linear(H, H * C).to(dtype)
        with torch.no_grad():
            bound = 1 / H ** 0.5
            self.linear.weight.copy_((torch.rand(H * C, H, generator=gen, dtype=torch.float64) * 2 - 1) * bound * scale)
            self.linear.bias.copy_((torch.rand(H * C, generator=gen, dtype=torch.float64) * 2 - 1) * bound * scale)

    def forward(self, t, z):
        return self.linear(z).view(*z.shape[:-1], self.H, self.C)


def cdeint_cases(ref):
    from oracle import cde as oracle_cde, interp
    gen = torch.Generator().manual_seed(777)
    cases = []
    specs = [
        # name, B, L, C, H, dtype, scale, explicit_t, method, options, t_out kind
        ("readme_toy_rk4", 1, 10, 2, 3, torch.float32, 1.0, "linspace01", "rk4", dict(step_size=1.0), "interval"),
        ("readme_toy_dopri5", 1, 10, 2, 3, torch.float32, 1.0, "linspace01", None, None, "interval"),
        ("mid_rk4_f32", 8, 16, 8, 32, torch.float32, 0.25, None, "rk4", dict(step_size=1.0), "interval"),
        ("mid_rk4_f64", 8, 16, 8, 32, torch.float64, 0.25, None, "rk4", dict(step_size=1.0), "interval"),
        ("mid_rk4_halfstep_multi_out", 5, 12, 8, 32, torch.float32, 0.25, None, "rk4", dict(step_size=0.5), "multi"),
        ("odd_dims_irregular_t", 3, 9, 3, 5, torch.float64, 0.5, "irregular", "rk4", dict(step_size=0.7), "multi"),
        ("mid_dopri5_f32", 6, 12, 4, 8, torch.float32, 0.25, None, "dopri5", None, "interval"),
    ]
    for name, B, L, C, H, dtype, scale, tkind, method, options, outkind in specs:
        if tkind == "linspace01":
            t = torch.linspace(0, 1, L, dtype=dtype)
            x = torch.cat([t.unsqueeze(0).unsqueeze(-1).expand(B, L, 1),
                           torch.rand(B, L, C - 1, generator=gen, dtype=dtype)], dim=2)
            coeffs = ref.hermite_cubic_coefficients_with_backward_differences(x)   # README passes no t
            knots_arg = None
        elif tkind == "irregular":
            t = _irregular_t(L, dtype, gen)
            x = torch.randn(B, L, C, generator=gen, dtype=dtype)
            coeffs = ref.hermite_cubic_coefficients_with_backward_differences(x, t)
            knots_arg = t
        else:
            x = 0.5 * torch.randn(B, L, C, generator=gen, dtype=dtype)
            x[..., 0] = torch.linspace(0, 1, L, dtype=dtype)
            coeffs = ref.hermite_cubic_coefficients_with_backward_differences(x)
            knots_arg = None
        func = LinearField(H, C, dtype, scale, gen)
        z0 = torch.randn(B, H, generator=gen, dtype=dtype)
        X = ref.CubicSpline(coeffs, knots_arg)
        if outkind == "interval":
            t_out = X.interval
        else:
            lo, hi = X.interval
            mids = torch.rand(4, generator=gen, dtype=torch.float64).sort().values.to(dtype) * (hi - lo) + lo
            t_out = torch.cat([lo.unsqueeze(0), mids, hi.unsqueeze(0)])
        kwargs = {}
        if method is not None:
            kwargs["method"] = method
        if options is not None:
            kwargs["options"] = options
        record = dict(name=name, coeffs=coeffs, knots=knots_arg, W=func.linear.weight.detach().clone(),
                      b=func.linear.bias.detach().clone(), z0=z0, t_out=t_out, method=method, options=options,
                      H=H, C=C)
        for adjoint in (False, True):
            z0g = z0.clone().requires_grad_(True)
            func.zero_grad()
            out = ref.cdeint(X=X, func=func, z0=z0g, t=t_out, adjoint=adjoint, **kwargs)
            assert out.shape == (B, len(t_out), H)
            weight = torch.linspace(0.5, 1.5, out.numel(), dtype=dtype).view_as(out)
            (out * weight).sum().backward()
            tag = "adjoint" if adjoint else "direct"
            record["out_" + tag] = out.detach().clone()
            record["gz0_" + tag] = z0g.grad.clone()
            record["gW_" + tag] = func.linear.weight.grad.clone()
            record["gb_" + tag] = func.linear.bias.grad.clone()
            # oracle.cde must reproduce the reference's solver.py plumbing bit-for-bit
            z0o = z0.clone().requires_grad_(True)
            func.zero_grad()
            out_o = oracle_cde.cdeint(interp.CubicPath(coeffs, knots_arg), func, z0o, t_out, adjoint=adjoint, **kwargs)
            (out_o * weight).sum().backward()
            assert torch.equal(out_o, out), name
            assert torch.equal(z0o.grad, record["gz0_" + tag]), name
            assert torch.equal(func.linear.weight.grad, record["gW_" + tag]), name
        record["loss_weight"] = "linspace(0.5, 1.5, numel)"
        cases.append(record)
    return cases


def gradient_cases(ref):
    """Gradients autograd produces through the REFERENCE's own construction code (test/test_tricks.py:21-49
    differentiates through it): fits with and without missing values, both fills, rectilinear preparation, spline
    evaluation w.r.t. coefficients / knot times / query times, and the log-ODE windows (the reference's windowing over
    oracle.logsig).  Every case: inputs, the loss weights `w` (loss = sum(out * w)), and the gradients."""
    gen = torch.Generator().manual_seed(77001)
    cases = []

    def gaps(x, p):
        x = x.masked_fill(torch.rand(x.shape, generator=gen) < p, float("nan"))
        flat = x.view(-1, x.size(-2), x.size(-1))
        flat[0, :, 0] = float("nan")
        flat[-1, :2, -1] = float("nan")
        flat[-1, -2:, -1] = float("nan")
        if flat.size(0) > 2:
            flat[1, :, 0] = float("nan")
            flat[1, x.size(-2) // 2, 0] = 0.25          # a single observation
        return x

    def record(kind, fn, x, t, want_t, **extra):
        xg = x.clone().requires_grad_(True)
        tg = None if t is None else t.clone().requires_grad_(want_t)
        out = fn(xg, tg)
        w = torch.randn(out.shape, generator=gen, dtype=out.dtype)
        (torch.nan_to_num(out) * w).sum().backward()
        cases.append(dict(kind=kind, x=x, t=t, w=w, out=out.detach(), grad_x=xg.grad,
                          grad_t=tg.grad if want_t else None, **extra))

    for dtype in (torch.float64, torch.float32):
        for batch, L, C in (((3,), 9, 2), ((2, 2), 14, 3), ((4,), 2, 2), ((4,), 3, 1)):
            x = torch.randn(*batch, L, C, generator=gen, dtype=dtype)
            t = _irregular_t(L, dtype, gen)
            record("hermite", ref.hermite_cubic_coefficients_with_backward_differences, x, t, True)
            record("natural", ref.natural_cubic_coeffs, x, t, True)
            xn = gaps(x.clone(), 0.3)
            record("hermite_nan", ref.hermite_cubic_coefficients_with_backward_differences, xn, t, False)
            record("natural_nan", ref.natural_cubic_coeffs, xn, t, False)
            record("natural_v0_nan", ref.natural_cubic_spline_coeffs, xn, t, False)
            record("linear_nan", ref.linear_interpolation_coeffs, xn, t, False)
            record("forward_fill", lambda a, _: ref.misc.forward_fill(a), xn, None, False)
            xr = xn.clone()
            xr[..., 0] = torch.arange(L, dtype=dtype) * 0.5
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                record("rectilinear", lambda a, _: ref.linear_interpolation_coeffs(a, rectilinear=0), xr, None, False)
        # evaluation of both controls: d/d(coefficients), d/d(knot times), d/d(query times)
        x = torch.randn(3, 11, 2, generator=gen, dtype=dtype)
        t = _irregular_t(11, dtype, gen)
        tq = _query_times(t, gen)
        for kind, make, coeffs in (("cubic", ref.CubicSpline, ref.natural_cubic_coeffs(x, t)),
                                   ("linear", ref.LinearInterpolation, x)):
            for what in ("evaluate", "derivative"):
                cg, tg, qg = coeffs.clone().requires_grad_(True), t.clone().requires_grad_(True), tq.clone().requires_grad_(True)
                out = getattr(make(cg, tg), what)(qg)
                w = torch.randn(out.shape, generator=gen, dtype=dtype)
                (out * w).sum().backward()
                cases.append(dict(kind="eval_" + kind + "_" + what, coeffs=coeffs, t=t, tq=tq, w=w, out=out.detach(),
                                  grad_coeffs=cg.grad, grad_t=tg.grad, grad_tq=qg.grad))
        # log-ODE windows (window ends between observations; gaps in the data)
        for C, depth, window in ((3, 3, 2.6), (2, 4, 4.0), (5, 2, 3.3)):
            x = (torch.randn(3, 17, C, generator=gen, dtype=dtype) * 0.4).cumsum(1)
            x[1, 4:7, 0] = float("nan")
            t = _irregular_t(17, dtype, gen)
            record("logsig", lambda a, tt, d=depth, wl=window: ref.logsig_windows(a, d, wl, tt), x, t, False,
                   depth=depth, window_length=window)
    return cases


def run_reference_tests():
    import pytest
    files = ["test_hermite_cubic.py", "test_natural_cubic_spline.py", "test_linear_interpolation.py", "test_misc.py",
             "test_cdeint.py", "test_tricks.py", "test_log_ode.py"]
    args = [os.path.join(REFERENCE, "test", f) for f in files]
    # torchsde-backed parametrisations cannot run (package absent): deselect them by keyword
    return pytest.main(args + ["-q", "-p", "no:cacheprovider", "-k", "not torchsde and not test_backend",
                               "--rootdir", "/tmp", "-c", "/dev/null"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run-reference-tests", action="store_true")
    ap.add_argument("--only-gradients", action="store_true", help="write gradients.pt and leave the other fixtures alone")
    ap.add_argument("--only-rebuilt-t", action="store_true", help="write logsig_rebuilt_t.pt and leave the others alone")
    opts = ap.parse_args()
    ref = import_reference()
    os.makedirs(OUT, exist_ok=True)
    rebuilt = logsig_rebuilt_t_cases(ref)
    torch.save(rebuilt, os.path.join(OUT, "logsig_rebuilt_t.pt"))
    print("logsig_rebuilt_t.pt: %d cases (same shape, different time grids; reference windowing over oracle.logsig)"
          % len(rebuilt))
    if opts.only_rebuilt_t:
        return
    grad_cases = gradient_cases(ref)
    torch.save(grad_cases, os.path.join(OUT, "gradients.pt"))
    print("gradients.pt: %d cases (autograd through the reference's own code)" % len(grad_cases))
    if opts.only_gradients:
        return
    interp_cases = interpolation_cases(ref)
    torch.save(interp_cases, os.path.join(OUT, "interpolation.pt"))
    print("interpolation.pt: %d cases (oracle bit-identical to reference on all)" % len(interp_cases))
    nan_cases = nan_fill_cases(ref)
    torch.save(nan_cases, os.path.join(OUT, "nan_fill.pt"))
    print("nan_fill.pt: %d cases (oracle bit-identical to reference on all)" % len(nan_cases))
    nat_cases = natural_cases(ref)
    torch.save(nat_cases, os.path.join(OUT, "natural_cubic.pt"))
    print("natural_cubic.pt: %d cases (oracle bit-identical to reference on all)" % len(nat_cases))
    ls_cases = logsig_cases(ref)
    torch.save(ls_cases, os.path.join(OUT, "logsig_windows.pt"))
    print("logsig_windows.pt: %d cases (oracle windowing bit-identical to the reference's, both over oracle.logsig)"
          % len(ls_cases))
    rect_cases = rectilinear_cases(ref)
    torch.save(rect_cases, os.path.join(OUT, "rectilinear.pt"))
    print("rectilinear.pt: %d cases (oracle bit-identical to reference on all; known-answer case of the reference's "
          "test included)" % len(rect_cases))
    cde_cases = cdeint_cases(ref)
    torch.save(cde_cases, os.path.join(OUT, "cdeint.pt"))
    print("cdeint.pt: %d cases (oracle.cde bit-identical to reference solver.py over oracle.odeint)" % len(cde_cases))
    if opts.run_reference_tests:
        sys.exit(run_reference_tests())


if __name__ == "__main__":
    main()
