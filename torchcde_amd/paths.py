"""Control paths: coefficient construction and the piecewise-polynomial path modules.

Host-side mirror (same names, argument meaning and error behaviour) of the reference's
    hermite_cubic_coefficients_with_backward_differences  interpolation_hermite_cubic_bdiff.py:23-44
    linear_interpolation_coeffs                           interpolation_linear.py:131-171
    CubicSpline / LinearInterpolation                     interpolation_cubic.py:268-336 / interpolation_linear.py:174-225
    InterpolationBase                                     interpolation_base.py:5-22
with every tensor operation executed by the HIP kernels of libcde_mi355x.so (K1, K1b).
"""
import abc
import math
import threading
import warnings

import torch

from . import _lib


# --------------------------------------------------------------------------------------- validation
def _validate_input_path(x, t):
    # error messages follow reference misc.py:70-100
    if not x.is_floating_point():
        raise ValueError("X must both be floating point.")
    if x.ndimension() < 2:
        raise ValueError("X must have at least two dimensions, corresponding to time and channels. It instead has "
                         "shape {}.".format(tuple(x.shape)))
    generated = t is None
    if generated:
        t = torch.linspace(0, x.size(-2) - 1, x.size(-2), dtype=x.dtype, device=x.device)
    if not t.is_floating_point():
        raise ValueError("t must both be floating point.")
    if len(t.shape) != 1:
        raise ValueError("t must be one dimensional. It instead has shape {}.".format(tuple(t.shape)))
    if not generated:
        # one device->host copy instead of the reference's per-element loop (misc.py:85-89)
        previous = -math.inf
        for value in t.detach().cpu().tolist():
            if value <= previous:
                raise ValueError("t must be monotonically increasing.")
            previous = value
    if x.size(-2) != t.size(0):
        raise ValueError("The time dimension of X must equal the length of t. X has shape {} and t has shape {}, "
                         "corresponding to time dimensions of {} and {} respectively."
                         .format(tuple(x.shape), tuple(t.shape), x.size(-2), t.size(0)))
    if t.size(0) < 2:
        raise ValueError("Must have a time dimension of size at least 2. It instead has shape {}, corresponding to a "
                         "time dimension of size {}.".format(tuple(t.shape), t.size(0)))
    return t


def _no_grad_through_path(*tensors):
    if torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in tensors):
        raise NotImplementedError(
            "torchcde_amd: gradients with respect to the knot times are not implemented for this construction "
            "(they are for the Hermite and natural cubic fits of data without missing values and for spline "
            "evaluation). Detach `t`.")


def _flat3(x):
    L, C = x.size(-2), x.size(-1)
    return x.detach().contiguous(), x.numel() // max(L * C, 1), L, C


def forward_fill(x, fill_index=-2):
    """Forward fill along ``fill_index`` (reference misc.py:103-126): every NaN takes the latest earlier
    observation, leading NaNs stay.  Returns ``x`` itself when nothing is missing, like the reference.  K0b."""
    assert isinstance(x, torch.Tensor)
    assert x.dim() >= 2
    _lib.require_gpu(x, "x")
    if not torch.isnan(x).any():
        return x
    dim = fill_index % x.dim()
    moved = x.movedim(dim, -2) if dim != x.dim() - 2 else x
    out = _CopyFill.apply(moved, None)
    return out.movedim(-2, dim) if dim != x.dim() - 2 else out


class _CopyFill(torch.autograd.Function):
    """K0b (``time_index is None``) / K0c with their backward: outputs are copies of input entries, the gradients of
    the copies flow back to the entry they were taken from (the reference's gathers, misc.py:103-126 and
    interpolation_linear.py:86-128, are differentiable)."""

    @staticmethod
    def forward(ctx, x, time_index):
        src, B, L, C = _flat3(x)
        lib = _lib.load()
        if time_index is None:
            out = torch.empty_like(src)
            _lib.check(lib.cde_forward_fill(_lib.ptr(src), _lib.ptr(out), B, L, C, _lib.dtype_enum(x.dtype),
                                            _lib.stream_ptr(x.device)), "cde_forward_fill")
        else:
            out = torch.empty(*x.shape[:-2], 2 * L - 1, C, dtype=x.dtype, device=x.device)
            _lib.check(lib.cde_rectilinear_prepare(_lib.ptr(src), _lib.ptr(out), B, L, C, time_index,
                                                   _lib.dtype_enum(x.dtype), _lib.stream_ptr(x.device)),
                       "cde_rectilinear_prepare")
        ctx.save_for_backward(src)
        ctx.time_index = time_index
        return out

    @staticmethod
    def backward(ctx, grad_out):
        src, = ctx.saved_tensors
        L, C = src.size(-2), src.size(-1)
        B = src.numel() // max(L * C, 1)
        grad_out = grad_out.contiguous()
        grad_x = torch.empty_like(src)
        lib = _lib.load()
        dt, stream = _lib.dtype_enum(src.dtype), _lib.stream_ptr(src.device)
        if ctx.time_index is None:
            _lib.check(lib.cde_forward_fill_backward(_lib.ptr(grad_out), _lib.ptr(src), _lib.ptr(grad_x), B, L, C, dt,
                                                     stream), "cde_forward_fill_backward")
        else:
            _lib.check(lib.cde_rectilinear_prepare_backward(_lib.ptr(grad_out), _lib.ptr(src), _lib.ptr(grad_x), B, L, C,
                                                            ctx.time_index, dt, stream),
                       "cde_rectilinear_prepare_backward")
        return grad_x, None


_RECTILINEAR_WARNING = ("The data `x` begins with missing values in some channels. The path will be constructed by "
                        "backward-filling the first observed value, which is not causal. Raising a warning as the "
                        "`rectilinear` argument has also been passed, which is nearly always only used when "
                        "causality is desired. If you need causality then fill in the missing value at the start of "
                        "each channel with whatever you'd like it to be. (The mean over that channel is a common "
                        "choice.)")


def _prepare_rectilinear_interpolation(data, time_index):
    """reference interpolation_linear.py:86-128 on the GPU (K0c): (..., L, C) -> (..., 2L-1, C)."""
    n_channels = data.size(-1)
    assert isinstance(time_index, int), "Index of the time channel must be an integer in [0, {}]".format(n_channels - 1)
    assert 0 <= time_index < n_channels, "Time index must be in [0, {}], was given {}." \
                                         "".format(n_channels - 1, time_index)
    _lib.require_gpu(data, "x")
    assert not torch.isnan(data[..., time_index]).any(), \
        "There exist nan values in the time column which is not allowed. If the times are padded with nans after " \
        "final time, a simple solution is to forward fill the final time."
    return _CopyFill.apply(data, time_index)


def linear_interpolation_coeffs(x, t=None, rectilinear=None):
    """Knots of the piecewise-linear control (reference interpolation_linear.py:131-171).

    Without missing values ``x`` itself is returned (same tensor object), exactly like the reference.  With NaNs
    every scalar path is filled by K0 (``cde_linear_fill_missing``): observed values stay, gaps become straight
    lines between the nearest observed neighbours, leading/trailing gaps are constant, all-NaN paths are zero.
    ``rectilinear=time_channel`` first applies the rectilinear preparation (K0c), as the reference does."""
    if rectilinear is not None:
        if x.dim() >= 2 and torch.isnan(x[..., 0, :]).any():
            warnings.warn(_RECTILINEAR_WARNING)
        x = _prepare_rectilinear_interpolation(x, rectilinear)
    t = _validate_input_path(x, t)
    _lib.require_gpu(x, "x")
    if not torch.isnan(x).any():
        return x
    _no_grad_through_path(t)
    knots = t.detach().to(device=x.device, dtype=x.dtype).contiguous()
    return _LinearFill.apply(x, knots)


class _LinearFill(torch.autograd.Function):
    """K0 with its backward: the gradient of every filled entry goes to the two observations it was interpolated from
    (``test/test_tricks.py:21-49`` differentiates through the coefficient construction)."""

    @staticmethod
    def forward(ctx, x, knots):
        src = x.detach().contiguous()
        out = torch.empty_like(src)
        L, C = src.size(-2), src.size(-1)
        B = src.numel() // (L * C)
        _lib.check(_lib.load().cde_linear_fill_missing(_lib.ptr(src), _lib.ptr(knots), _lib.ptr(out), B, L, C,
                                                       _lib.dtype_enum(x.dtype), _lib.stream_ptr(x.device)),
                   "cde_linear_fill_missing")
        ctx.save_for_backward(src, knots)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        src, knots = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_x = torch.empty_like(src)
        L, C = src.size(-2), src.size(-1)
        B = src.numel() // (L * C)
        _lib.check(_lib.load().cde_linear_fill_missing_backward(
            _lib.ptr(grad_out), _lib.ptr(src), _lib.ptr(knots), _lib.ptr(grad_x), B, L, C, _lib.dtype_enum(src.dtype),
            _lib.stream_ptr(src.device)), "cde_linear_fill_missing_backward")
        return grad_x, None


def _natural_cubic(x, t, version):
    t = _validate_input_path(x, t)
    _lib.require_gpu(x, "x")
    has_missing = bool(torch.isnan(x).any())
    if torch.is_grad_enabled() and (x.requires_grad or t.requires_grad):
        if has_missing:
            if t.requires_grad:
                raise NotImplementedError("torchcde_amd: gradients of the natural cubic fit w.r.t. the knot times are "
                                          "implemented for data without missing values only.")
            return _NaturalCubicFitMissing.apply(x, t, version)
        return _NaturalCubicFit.apply(x, t, version)
    knots = t.detach().to(device=x.device, dtype=x.dtype).contiguous()
    return _natural_cubic_forward(x, knots, version, has_missing)


def _natural_cubic_forward(x, knots, version, has_missing):
    src, B, L, C = _flat3(x)
    out = torch.empty(*x.shape[:-2], L - 1, 4 * C, dtype=x.dtype, device=x.device)
    lib = _lib.load()
    _lib.check(lib.cde_natural_cubic_coeffs(_lib.ptr(src), _lib.ptr(knots), _lib.ptr(out), B, L, C, version,
                                            int(has_missing), _lib.dtype_enum(x.dtype),
                                            _lib.stream_ptr(x.device)), "cde_natural_cubic_coeffs")
    return out


class _NaturalCubicFit(torch.autograd.Function):
    """K1n with its backward (data without missing values): the coefficients are linear in the values -- the gradient
    is one more solve with the same tridiagonal matrix -- and smooth in the knot times
    (``cde_natural_cubic_coeffs_backward``; reference test/test_tricks.py:21-49 differentiates w.r.t. both)."""

    @staticmethod
    def forward(ctx, x, t, version):
        knots = t.detach().to(device=x.device, dtype=x.dtype).contiguous()
        src = x.detach().contiguous()
        ctx.save_for_backward(knots, src)
        ctx.t_meta = (t.dtype, t.device)
        return _natural_cubic_forward(x, knots, version, False)

    @staticmethod
    def backward(ctx, grad):
        knots, src = ctx.saved_tensors
        shape = src.shape
        L, C = shape[-2], shape[-1]
        B = src.numel() // (L * C)
        grad = grad.contiguous()
        grad_x = torch.empty(shape, dtype=grad.dtype, device=grad.device)
        lib = _lib.load()
        dt = _lib.dtype_enum(grad.dtype)
        nbytes = lib.cde_natural_cubic_coeffs_backward_workspace_bytes(L, dt)
        workspace = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=grad.device)
        want_t = ctx.needs_input_grad[1]
        kd = torch.empty_like(src) if want_t else None
        rows = torch.empty_like(src) if want_t else None
        _lib.check(lib.cde_natural_cubic_coeffs_backward(
            _lib.ptr(grad), _lib.ptr(knots), _lib.ptr(grad_x), _lib.ptr(workspace), nbytes, B, L, C, dt,
            _lib.ptr(src) if want_t else None, _lib.ptr(kd), _lib.ptr(rows), _lib.stream_ptr(grad.device)),
            "cde_natural_cubic_coeffs_backward")
        grad_t = None
        if want_t:
            t_dtype, t_device = ctx.t_meta
            grad_t = rows.reshape(-1, L, C).sum(dim=(0, 2)).to(device=t_device, dtype=t_dtype)
        return (grad_x if ctx.needs_input_grad[0] else None), grad_t, None


class _NaturalCubicFitMissing(torch.autograd.Function):
    """K1n on data with missing values, gradient w.r.t. the observed values
    (``cde_natural_cubic_coeffs_backward_missing``; autograd through interpolation_cubic.py:83-166)."""

    @staticmethod
    def forward(ctx, x, t, version):
        knots = t.detach().to(device=x.device, dtype=x.dtype).contiguous()
        src = x.detach().contiguous()
        ctx.save_for_backward(knots, src)
        ctx.version = version
        return _natural_cubic_forward(x, knots, version, True)

    @staticmethod
    def backward(ctx, grad):
        knots, src = ctx.saved_tensors
        L, C = src.shape[-2], src.shape[-1]
        B = src.numel() // (L * C)
        grad = grad.contiguous()
        grad_x = torch.empty_like(src)
        workspace = torch.empty_like(grad)
        _lib.check(_lib.load().cde_natural_cubic_coeffs_backward_missing(
            _lib.ptr(grad), _lib.ptr(src), _lib.ptr(knots), _lib.ptr(grad_x), _lib.ptr(workspace), B, L, C, ctx.version,
            _lib.dtype_enum(grad.dtype), _lib.stream_ptr(grad.device)), "cde_natural_cubic_coeffs_backward_missing")
        return grad_x, None, None


def natural_cubic_coeffs(x, t=None):
    """Natural cubic spline coefficients (reference interpolation_cubic.py:226-266), missing values (NaN) supported;
    pass the result to ``CubicSpline``.  K1n: one kernel, bit-exact."""
    return _natural_cubic(x, t, 1)


def natural_cubic_spline_coeffs(x, t=None):
    """Deprecated variant kept by the reference (interpolation_cubic.py:186-223): differs from ``natural_cubic_coeffs``
    only in how missing values at the two ends of a series are imputed."""
    return _natural_cubic(x, t, 0)


def _path_eval(coeffs, knots, flat, n_intervals, C, degree, what):
    out = torch.empty(coeffs.size(0), flat.numel(), C, dtype=coeffs.dtype, device=coeffs.device)
    lib = _lib.load()
    _lib.check(lib.cde_path_eval(_lib.ptr(coeffs), _lib.ptr(knots), _lib.ptr(flat), flat.numel(), _lib.ptr(out),
                                 coeffs.size(0), n_intervals, C, degree, what, _lib.dtype_enum(coeffs.dtype),
                                 _lib.stream_ptr(coeffs.device)), "cde_path_eval")
    return out


class _PathEval(torch.autograd.Function):
    """evaluate / derivative as a differentiable function of the path's coefficient buffers (cubic: a, b, 2c, 3d;
    linear: the knot values), of the query times and, for a cubic spline, of its knot times (both through
    ``frac = t - t_i``, interpolation_cubic.py:315-336; reference test/test_tricks.py:21-49 asks for these gradients)."""

    @staticmethod
    def forward(ctx, coeffs, knots, flat, n_intervals, C, degree, what, knot_times, query, *pieces):
        ctx.save_for_backward(knots, flat, coeffs)
        ctx.query_shape = None if query is None else tuple(query.shape)
        ctx.meta = (tuple(coeffs.shape), n_intervals, C, degree, what, [tuple(p.shape) for p in pieces])
        return _path_eval(coeffs, knots, flat, n_intervals, C, degree, what)

    @staticmethod
    def backward(ctx, grad_out):
        knots, flat, coeffs = ctx.saved_tensors
        flat_shape, n_intervals, C, degree, what, shapes = ctx.meta
        g = grad_out.contiguous()
        lib = _lib.load()
        parts = (None,) * len(shapes)
        if any(ctx.needs_input_grad[9:]):
            grad = torch.zeros(flat_shape, dtype=g.dtype, device=g.device)
            _lib.check(lib.cde_path_eval_backward(_lib.ptr(g), _lib.ptr(knots), _lib.ptr(flat), flat.numel(), _lib.ptr(grad),
                                                  flat_shape[0], n_intervals, C, degree, what, _lib.dtype_enum(g.dtype),
                                                  _lib.stream_ptr(g.device)), "cde_path_eval_backward")
            if len(shapes) == 1:
                parts = (grad.reshape(shapes[0]),)
            else:
                parts = tuple(grad[..., k * C:(k + 1) * C].reshape(shape) for k, shape in enumerate(shapes))
        grad_knots = grad_query = None
        if ctx.needs_input_grad[7] or ctx.needs_input_grad[8]:
            # d out / d frac at every query; frac = t - knot[index]: a query moves with its own time and against ONE knot
            nq = flat.numel()
            index = torch.empty(nq, dtype=torch.int64, device=g.device)
            frac = torch.empty(nq, dtype=g.dtype, device=g.device)
            _lib.check(lib.cde_interpret_t(_lib.ptr(knots), n_intervals, _lib.ptr(flat), nq, _lib.ptr(index),
                                           _lib.ptr(frac), _lib.dtype_enum(g.dtype), _lib.stream_ptr(g.device)),
                       "cde_interpret_t")
            fr = frac.reshape(1, nq, 1)
            if degree == _lib.PATH_CUBIC:
                rows = coeffs[:, index]                                          # (B, nq, 4C)
                b, two_c, three_d = rows[..., C:2 * C], rows[..., 2 * C:3 * C], rows[..., 3 * C:]
                dfrac = two_c + 2 * three_d * fr if what == _lib.EVAL_DERIVATIVE else b + (two_c + three_d * fr) * fr
            else:
                width = (knots[index + 1] - knots[index]).reshape(1, nq, 1)
                slope = (coeffs[:, index + 1] - coeffs[:, index]) / width
                dfrac = torch.zeros_like(slope) if what == _lib.EVAL_DERIVATIVE else slope
            gq = g.reshape(-1, nq, C)
            per_query = (gq * dfrac).sum(dim=(0, 2))
            if ctx.needs_input_grad[7]:
                grad_knots = torch.zeros(n_intervals + 1, dtype=g.dtype, device=g.device).index_add_(0, index, -per_query)
                if degree != _lib.PATH_CUBIC:
                    # piecewise linear (interpolation_linear.py:212-225): the slope (x_{i+1} - x_i) / (t_{i+1} - t_i)
                    # also moves with both knots of its interval
                    scale = 1 / width if what == _lib.EVAL_DERIVATIVE else fr / width
                    widen = (gq * slope * scale).sum(dim=(0, 2))
                    grad_knots.index_add_(0, index, widen).index_add_(0, index + 1, -widen)
            if ctx.needs_input_grad[8]:
                grad_query = per_query.reshape(ctx.query_shape)
        return ((None,) * 7 + (grad_knots, grad_query)
                + tuple(p if need else None for p, need in zip(parts, ctx.needs_input_grad[9:])))


class _HermiteFit(torch.autograd.Function):
    """K1 with its transpose as the backward (the fit is linear in x) and, when the knot times require a gradient, the
    derivative w.r.t. the interval widths (``cde_hermite_bdiff_coeffs_backward_dt``)."""

    @staticmethod
    def forward(ctx, x, t):
        L, C = x.size(-2), x.size(-1)
        batch = x.shape[:-2]
        knots = t.detach().to(device=x.device, dtype=x.dtype).contiguous()
        src, B, _, _ = _flat3(x)
        out = torch.empty(*batch, L - 1, 4 * C, dtype=x.dtype, device=x.device)
        lib = _lib.load()
        _lib.check(lib.cde_hermite_bdiff_coeffs(_lib.ptr(src), _lib.ptr(knots), _lib.ptr(out), B, L, C,
                                                _lib.dtype_enum(x.dtype), _lib.stream_ptr(x.device)),
                   "cde_hermite_bdiff_coeffs")
        ctx.save_for_backward(knots, src if t.requires_grad else None)
        ctx.shape = tuple(x.shape)
        ctx.t_meta = (t.dtype, t.device)
        return out

    @staticmethod
    def backward(ctx, grad_coeffs):
        knots, src = ctx.saved_tensors
        L, C = ctx.shape[-2], ctx.shape[-1]
        g = grad_coeffs.contiguous()
        B = g.numel() // ((L - 1) * 4 * C)
        lib = _lib.load()
        grad_x = grad_t = None
        if ctx.needs_input_grad[0]:
            grad_x = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
            _lib.check(lib.cde_hermite_bdiff_coeffs_backward(_lib.ptr(g), _lib.ptr(knots), _lib.ptr(grad_x), B, L, C,
                                                             _lib.dtype_enum(g.dtype), _lib.stream_ptr(g.device)),
                       "cde_hermite_bdiff_coeffs_backward")
        if ctx.needs_input_grad[1]:
            gh = torch.empty(B, L - 1, C, dtype=g.dtype, device=g.device)
            _lib.check(lib.cde_hermite_bdiff_coeffs_backward_dt(_lib.ptr(g), _lib.ptr(src), _lib.ptr(knots), _lib.ptr(gh),
                                                                B, L, C, _lib.dtype_enum(g.dtype),
                                                                _lib.stream_ptr(g.device)),
                       "cde_hermite_bdiff_coeffs_backward_dt")
            per_interval = gh.sum(dim=(0, 2))
            grad_t = torch.zeros(L, dtype=g.dtype, device=g.device)
            grad_t[1:] += per_interval
            grad_t[:-1] -= per_interval
            grad_t = grad_t.to(device=ctx.t_meta[1], dtype=ctx.t_meta[0])
        return grad_x, grad_t


def hermite_cubic_coefficients_with_backward_differences(x, t=None):
    """Hermite cubic spline coefficients with backward differences, (..., L-1, 4C) = [a | b | 2c | 3d].

    Same contract as reference interpolation_hermite_cubic_bdiff.py:23-44; computed by K1
    (``cde_hermite_bdiff_coeffs``) in one pass over ``x``.  Differentiable w.r.t. ``x`` (data without missing
    values) and w.r.t. ``t`` (data without missing values), like the reference's eager ops."""
    t_grad = torch.is_grad_enabled() and isinstance(t, torch.Tensor) and t.requires_grad
    if t_grad or (torch.is_grad_enabled() and x.requires_grad):
        if t_grad and bool(torch.isnan(x).any()):
            raise NotImplementedError("torchcde_amd: gradients with respect to the knot times through the fit are "
                                      "implemented for data without missing values only.")
        filled = x if t_grad else linear_interpolation_coeffs(x, t=t, rectilinear=None)
        if t is None:
            t = torch.linspace(0, filled.size(-2) - 1, filled.size(-2), dtype=filled.dtype, device=filled.device)
        else:
            _validate_input_path(filled, t)
        return _HermiteFit.apply(filled, t)
    # No gradient wanted: the reference's NaN scan (linear_interpolation_coeffs, interpolation_linear.py:169) rides on
    # the fit itself -- K1 reads every value anyway -- and its consequence is drawn ON THE DEVICE: the checked fit raises a
    # per-stream flag to this call's number, and two launches gated by that flag fill the gaps (K0) and refit.  No
    # read-back, no host sync (round 3 read 4 bytes back per call: 141 us per call around a 111 us kernel).
    knots = _validate_input_path(x, t)
    _lib.require_gpu(x, "x")
    _no_grad_through_path(t)
    knots = knots.detach().to(device=x.device, dtype=x.dtype).contiguous()
    L, C = x.size(-2), x.size(-1)
    batch = x.shape[:-2]
    src = x.detach().contiguous()
    B = src.numel() // (L * C)
    out = torch.empty(*batch, L - 1, 4 * C, dtype=x.dtype, device=x.device)
    # Where the filled series would go if the device-side check finds gaps (caching allocator: no sync).  It is the price of
    # not reading the flag back: peak memory of a no-grad fit is x + 4x (coefficients) + x (this buffer, freed on return) --
    # a dataset fitted in one call needs 6x its size where round 3 needed 5x; fit in chunks when that
    # matters, the per-call cost is 0.12 ms.
    scratch = torch.empty_like(src)
    lib = _lib.load()
    stream = _lib.stream_ptr(x.device)
    args = (_lib.ptr(src), _lib.ptr(knots), _lib.ptr(out), _lib.ptr(scratch), B, L, C, _lib.dtype_enum(x.dtype))
    if torch.cuda.is_current_stream_capturing():
        # a captured call is replayed with its arguments baked in, so a call number would be stale from the second replay
        # on: the graph gets a flag word of its own (from the graph's pool) that a captured fill kernel zeroes first
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(lib.cde_hermite_bdiff_coeffs_nonblocking(*args, _lib.ptr(flag), 1, stream), "cde_hermite_bdiff_coeffs_nonblocking")
        return out
    # the call number is drawn AND the three launches are queued under one lock: numbers then rise in stream order, which
    # is what the flag protocol assumes (two threads sharing a stream could queue call 5 after call 6; 5's
    # atomicMax then did nothing, its gate failed and its gaps stayed NaN)
    with _NAN_FLAGS_LOCK:
        flag, generation = _nan_flag(x.device, stream)
        rc = lib.cde_hermite_bdiff_coeffs_nonblocking(*args, _lib.ptr(flag), generation, stream)
    _lib.check(rc, "cde_hermite_bdiff_coeffs_nonblocking")
    return out


_NAN_FLAGS = {}
_NAN_FLAGS_LOCK = threading.Lock()


def _nan_flag(device, stream):
    """The device int K1 raises when its input holds NaNs, one per (device, stream), and the number of this call on it
    (cde_hermite_bdiff_coeffs_nonblocking: the flag is compared with the call's number, so it is never zeroed again).
    The caller holds _NAN_FLAGS_LOCK until its launches are queued."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(),
           getattr(stream, "value", stream))
    entry = _NAN_FLAGS.get(key)
    if entry is None or entry[1] >= 2 ** 31 - 2:
        entry = [torch.zeros(1, dtype=torch.int32, device=device), 0]
        _NAN_FLAGS[key] = entry
    entry[1] += 1
    return entry[0], entry[1]


# --------------------------------------------------------------------------------------- path modules
class InterpolationBase(torch.nn.Module, metaclass=abc.ABCMeta):
    """Contract every control satisfies (reference interpolation_base.py:5-22)."""

    @property
    @abc.abstractmethod
    def grid_points(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def interval(self):
        raise NotImplementedError

    @abc.abstractmethod
    def evaluate(self, t):
        raise NotImplementedError

    @abc.abstractmethod
    def derivative(self, t):
        raise NotImplementedError


class _NativePath(InterpolationBase):
    """Shared machinery: flatten batch dims, call K1b, restore the reference's output shape
    ``batch_dims + t.shape + (channels,)``."""

    _degree = None

    def _packed(self):
        raise NotImplementedError

    def _channels(self):
        raise NotImplementedError

    def _n_intervals(self):
        return self._t.size(0) - 1

    @property
    def grid_points(self):
        return self._t

    @property
    def interval(self):
        # One tensor object per (knots tensor, version): lets cdeint recognise repeated calls with
        # ``t=X.interval`` without reading the values back from the device every time.
        t = self._t
        cached = getattr(self, "_interval_cache", None)
        if cached is None or cached[0] is not t or cached[1] != t._version:
            cached = (t, t._version, torch.stack([t[0], t[-1]]))
            object.__setattr__(self, "_interval_cache", cached)
        return cached[2]

    def _native_inputs(self):
        """(coeffs flattened to (B, rows, width) contiguous, knots, batch_shape)"""
        coeffs = self._packed()
        _lib.require_gpu(coeffs, "the control path")
        knots = self._t
        if knots.dtype != coeffs.dtype or knots.device != coeffs.device:
            knots = knots.to(device=coeffs.device, dtype=coeffs.dtype)
        batch = coeffs.shape[:-2]
        flat = coeffs.detach().reshape(-1, coeffs.size(-2), coeffs.size(-1)).contiguous()
        return flat, knots.detach().contiguous(), batch

    def _interpret_t(self, t):
        """(fractional_part, index) exactly as the reference's ``_interpret_t`` (int64 index)."""
        coeffs, knots, _ = self._native_inputs()
        tq = torch.as_tensor(t, dtype=coeffs.dtype, device=coeffs.device)
        flat = tq.detach().reshape(-1).contiguous()
        index = torch.empty(flat.numel(), dtype=torch.int64, device=coeffs.device)
        frac = torch.empty(flat.numel(), dtype=coeffs.dtype, device=coeffs.device)
        lib = _lib.load()
        _lib.check(lib.cde_interpret_t(_lib.ptr(knots), self._n_intervals(), _lib.ptr(flat), flat.numel(),
                                       _lib.ptr(index), _lib.ptr(frac), _lib.dtype_enum(coeffs.dtype),
                                       _lib.stream_ptr(coeffs.device)), "cde_interpret_t")
        return frac.reshape(tq.shape), index.reshape(tq.shape)

    def _eval(self, t, what):
        coeffs, knots, batch = self._native_inputs()
        tq = torch.as_tensor(t, dtype=coeffs.dtype, device=coeffs.device)
        flat = tq.detach().reshape(-1).contiguous()
        C = self._channels()
        pieces = self._coefficient_buffers()
        grad_mode = torch.is_grad_enabled()
        knot_times = self._t if (grad_mode and self._t.requires_grad) else None
        query = tq if (grad_mode and tq.requires_grad) else None
        if grad_mode and (knot_times is not None or query is not None or any(p.requires_grad for p in pieces)):
            # differentiable w.r.t. the coefficients (like the reference's gathers): K1b forward, scatter kernel backward;
            # w.r.t. the query times and a cubic spline's knot times through frac = t - t_i.  The buffers themselves are
            # the autograd inputs (a view re-assembled with as_strided would only carry the gradient of its first block).
            out = _PathEval.apply(coeffs, knots, flat, self._n_intervals(), C, self._degree, what, knot_times, query,
                                  *pieces)
        else:
            out = _path_eval(coeffs, knots, flat, self._n_intervals(), C, self._degree, what)
        return out.reshape(*batch, *tq.shape, C)

    def evaluate(self, t):
        return self._eval(t, _lib.EVAL_VALUE)

    def derivative(self, t):
        return self._eval(t, _lib.EVAL_DERIVATIVE)

    def _second_derivative(self, t):
        """d/dt of ``derivative(t)`` for a scalar ``t`` (what autograd gives the reference through ``frac = t - t_i``,
        interpolation_cubic.py:331-336; zero for a piecewise-linear control, interpolation_linear.py:222-225).  Used only
        for gradients with respect to times; a few torch ops on the coefficient buffers, no gradient of its own."""
        raise NotImplementedError


class CubicSpline(_NativePath):
    """Piecewise-cubic control built from packed coefficients (reference interpolation_cubic.py:268-336).

    Buffers keep the reference's names (``_t, _a, _b, _two_c, _three_d``: views of the packed tensor)
    so ``state_dict`` / ``.to()`` / the requires-grad warning of ``cdeint`` behave the same."""

    _degree = _lib.PATH_CUBIC

    def __init__(self, coeffs, t=None, **kwargs):
        super().__init__(**kwargs)
        if t is None:
            t = torch.linspace(0, coeffs.size(-2), coeffs.size(-2) + 1, dtype=coeffs.dtype, device=coeffs.device)
        channels = coeffs.size(-1) // 4
        if channels * 4 != coeffs.size(-1):
            raise ValueError("Passed invalid coeffs.")
        self._C = channels
        self.register_buffer("_t", t)
        self.register_buffer("_a", coeffs[..., :channels])
        self.register_buffer("_b", coeffs[..., channels:2 * channels])
        self.register_buffer("_two_c", coeffs[..., 2 * channels:3 * channels])
        self.register_buffer("_three_d", coeffs[..., 3 * channels:])

    def _channels(self):
        return self._C

    def _control_buffers(self):
        """Buffers the control derivative reads (gradient targets for adjoint_params=(..., coeffs))."""
        return (self._b, self._two_c, self._three_d)

    def _second_derivative(self, t):
        with torch.no_grad():
            frac, index = self._interpret_t(t)
            index = index.reshape(())
            return self._two_c[..., index, :] + 2 * self._three_d[..., index, :] * frac.reshape(())

    def _coefficient_buffers(self):
        return (self._a, self._b, self._two_c, self._three_d)

    def _packed(self):
        a, b, c, d = self._a, self._b, self._two_c, self._three_d
        C = self._C
        # fast path: the four buffers are still adjacent views of one packed (..., L-1, 4C) tensor
        try:
            same = (a.untyped_storage().data_ptr() == d.untyped_storage().data_ptr()
                    and a.stride() == b.stride() == c.stride() == d.stride() and a.stride(-1) == 1
                    and a.stride(-2) == 4 * C
                    and b.storage_offset() == a.storage_offset() + C and c.storage_offset() == a.storage_offset() + 2 * C
                    and d.storage_offset() == a.storage_offset() + 3 * C)
        except RuntimeError:
            same = False
        if same:
            shape = tuple(a.shape[:-1]) + (4 * C,)
            return torch.as_strided(a, shape, a.stride(), a.storage_offset())
        return torch.cat([a, b, c, d], dim=-1)


class NaturalCubicSpline(CubicSpline):
    """Deprecated alias kept by the reference (interpolation_cubic.py:339-346)."""


class LinearInterpolation(_NativePath):
    """Piecewise-linear control (reference interpolation_linear.py:174-225)."""

    _degree = _lib.PATH_LINEAR

    def __init__(self, coeffs, t=None, **kwargs):
        super().__init__(**kwargs)
        if t is None:
            t = torch.linspace(0, coeffs.size(-2) - 1, coeffs.size(-2), dtype=coeffs.dtype, device=coeffs.device)
        self.register_buffer("_t", t)
        self.register_buffer("_coeffs", coeffs)

    def _channels(self):
        return self._coeffs.size(-1)

    def _control_buffers(self):
        return (self._coeffs,)

    def _coefficient_buffers(self):
        return (self._coeffs,)

    def _second_derivative(self, t):
        return torch.zeros_like(self._coeffs[..., 0, :])

    def _packed(self):
        return self._coeffs
