"""Oracle: control-path construction and evaluation (CPU, torch eager).  TEST INFRASTRUCTURE.

Restates, operation-for-operation (so that float results are bit-identical to
the reference on CPU), the interpolation half of the hot path:

  check_path                 <- torchcde/misc.py:70-100            (validate_input_path)
  forward_fill               <- torchcde/misc.py:103-126
  rectilinear_prepare        <- torchcde/interpolation_linear.py:86-128
  linear_coeffs              <- torchcde/interpolation_linear.py:131-171 (incl. the NaN fill :13-84 and rectilinear)
  natural_cubic_coeffs       <- torchcde/interpolation_cubic.py:7-266 + misc.py:14-67 (tridiagonal solve)
  hermite_bdiff_coeffs       <- torchcde/interpolation_hermite_cubic_bdiff.py:5-44
  locate                     <- torchcde/interpolation_cubic.py:315-322  (== interpolation_linear.py:203-210)
  cubic_value / cubic_slope  <- torchcde/interpolation_cubic.py:324-336
  linear_value / linear_slope<- torchcde/interpolation_linear.py:212-225
  CubicPath / LinearPath     <- the nn.Module shells (:282-313 / :177-201) so that
                                oracle.cde can drive them exactly as the reference does.

PINNED against the imported reference by oracle/make_golden.py (bitwise) and by
tests/test_oracle.py (golden fixtures + the closed-form unit-time Hermite of the
reference's test/test_hermite_cubic.py:6-21).
"""
import math

import torch


# ----------------------------------------------------------------------------- validation
def check_path(x, t):
    """misc.py:70-100.  Returns the (possibly defaulted) knot vector."""
    if not x.is_floating_point():
        raise ValueError("X must both be floating point.")
    if x.ndimension() < 2:
        raise ValueError("X must have at least two dimensions, corresponding to time and channels. It instead has "
                         "shape {}.".format(tuple(x.shape)))
    length = x.size(-2)
    if t is None:
        t = torch.linspace(0, length - 1, length, dtype=x.dtype, device=x.device)
    if not t.is_floating_point():
        raise ValueError("t must both be floating point.")
    if t.dim() != 1:
        raise ValueError("t must be one dimensional. It instead has shape {}.".format(tuple(t.shape)))
    last = -math.inf
    for value in t.tolist():
        if value <= last:
            raise ValueError("t must be monotonically increasing.")
        last = value
    if length != t.size(0):
        raise ValueError("The time dimension of X must equal the length of t. X has shape {} and t has shape {}, "
                         "corresponding to time dimensions of {} and {} respectively."
                         .format(tuple(x.shape), tuple(t.shape), length, t.size(0)))
    if t.size(0) < 2:
        raise ValueError("Must have a time dimension of size at least 2. It instead has shape {}, corresponding to a "
                         "time dimension of size {}.".format(tuple(t.shape), t.size(0)))
    return t


def _fill_scalar_path(t, x):
    """interpolation_linear.py:13-69 for one scalar path (length,): missing values (NaN) are replaced by the
    linear interpolant between the nearest OBSERVED neighbours; leading / trailing gaps take the first / last
    observation; an all-NaN path becomes zeros.  Same arithmetic: prev + ((t - t_prev)/(t_next - t_prev))*(next - prev)."""
    observed = ~torch.isnan(x)
    n_obs = int(observed.sum())
    if n_obs == 0:
        return torch.zeros_like(x)
    if n_obs == x.numel():
        return x
    x = x.clone()
    obs_idx = observed.nonzero().flatten().tolist()
    if not observed[0]:
        x[0] = x[obs_idx[0]]
    if not observed[-1]:
        x[-1] = x[obs_idx[-1]]
    anchors = sorted(set(obs_idx) | {0, x.numel() - 1})
    for lo, hi in zip(anchors[:-1], anchors[1:]):
        for i in range(lo + 1, hi):
            ratio = (t[i] - t[lo]) / (t[hi] - t[lo])
            x[i] = x[lo] + ratio * (x[hi] - x[lo])
    return x


def forward_fill(x):
    """misc.py:103-126 along the length axis (-2), as a plain loop: NaNs take the latest earlier observation of their
    scalar path; leading NaNs stay -- each one is its own entry (the reference's gather indexes a leading NaN to
    itself, which matters to autograd only)."""
    out = x.clone()
    seen = ~torch.isnan(x[..., 0, :])
    for i in range(1, x.size(-2)):
        cur = out[..., i, :]
        out[..., i, :] = torch.where(torch.isnan(cur) & seen, out[..., i - 1, :], cur)
        seen = seen | ~torch.isnan(cur)
    return out


def rectilinear_prepare(x, time_index):
    """interpolation_linear.py:86-128: forward fill, repeat every row twice, advance the time channel by one row,
    drop the last row -> (..., 2L-1, C)."""
    assert isinstance(time_index, int) and 0 <= time_index < x.size(-1)
    assert not torch.isnan(x[..., time_index]).any()
    filled = forward_fill(x)
    L = x.size(-2)
    rows = []
    for j in range(2 * L - 1):
        row = filled[..., j // 2, :].clone()
        row[..., time_index] = filled[..., (j + 1) // 2, time_index]
        rows.append(row)
    return torch.stack(rows, dim=-2)


def linear_coeffs(x, t=None, rectilinear=None):
    """interpolation_linear.py:131-171: optional rectilinear preparation, then ``x`` itself when nothing is missing,
    otherwise every scalar path (one per series and channel) filled by ``_fill_scalar_path`` (:72-84)."""
    if rectilinear is not None:
        x = rectilinear_prepare(x, rectilinear)
    t = check_path(x, t)
    if not torch.isnan(x).any():
        return x
    flat = x.transpose(-1, -2).reshape(-1, x.size(-2))
    filled = torch.stack([_fill_scalar_path(t, row) for row in flat])
    return filled.reshape(*x.shape[:-2], x.size(-1), x.size(-2)).transpose(-1, -2)


# ----------------------------------------------------------------------------- Hermite fit
def hermite_bdiff_coeffs(x, t=None):
    """interpolation_hermite_cubic_bdiff.py:23-44 + :5-20.

    Output (..., L-1, 4C) = [a | b | 2c | 3d] per interval.  The arithmetic below keeps
    the reference's association order: every intermediate is the same float.
    """
    knots_in = linear_coeffs(x, t)
    if t is None:
        t = torch.linspace(0, knots_in.size(-2) - 1, knots_in.size(-2), dtype=knots_in.dtype, device=knots_in.device)
    left = knots_in[..., :-1, :]
    right = knots_in[..., 1:, :]
    h = (t[1:] - t[:-1]).unsqueeze(-1)
    # secant slope on each interval (:39)
    secant = (right - left) / h
    # knot derivative entering each interval = slope of the interval before it; the
    # first interval re-uses its own slope (:10)
    enter = torch.cat((secant[..., [0], :], secant[..., :-1, :]), dim=-2)
    rise = right - left
    a = left
    b = enter
    two_c = 2 * (3 * (rise / h - b) - secant + enter) / h          # :17
    three_d = (1 / h ** 2) * (secant - b) - (two_c) / h            # :18
    return torch.cat([a, b, two_c, three_d], dim=-1)


# ----------------------------------------------------------------------------- natural cubic splines
def _natural_pieces(times, values):
    """interpolation_cubic.py:7-54 for ONE scalar path without missing values: (a, b, two_c, three_d) per piece, with
    the tridiagonal solve of misc.py:14-67 written out (same operation order)."""
    m = values.numel()
    if m == 2:
        zero = torch.zeros(1, dtype=values.dtype)
        return values[:1], (values[1:] - values[:1]) / (times[1:] - times[:1]), zero, zero.clone()
    r = (times[1:] - times[:-1]).reciprocal()
    r2 = r ** 2
    three = 3 * (values[1:] - values[:-1])
    six = 2 * three
    scaled = three * r2
    diag = torch.empty(m, dtype=values.dtype)
    diag[:-1] = r
    diag[-1] = 0
    diag[1:] += r
    diag *= 2
    rhs = torch.empty(m, dtype=values.dtype)
    rhs[:-1] = scaled
    rhs[-1] = 0
    rhs[1:] += scaled
    new_d, new_b = [diag[0]], [rhs[0]]
    for i in range(1, m):
        w = r[i - 1] / new_d[i - 1]
        new_d.append(diag[i] - w * r[i - 1])
        new_b.append(rhs[i] - w * new_b[i - 1])
    kd = [None] * m
    kd[m - 1] = new_b[m - 1] / new_d[m - 1]
    for i in range(m - 2, -1, -1):
        kd[i] = (new_b[i] - r[i] * kd[i + 1]) / new_d[i]
    kd = torch.stack(kd)
    a = values[:-1]
    b = kd[:-1]
    two_c = (six * r - 4 * kd[:-1] - 2 * kd[1:]) * r
    three_d = (-six * r + 3 * (kd[:-1] + kd[1:])) * r2
    return a, b, two_c, three_d


def _natural_scalar_path(t, x, version):
    """interpolation_cubic.py:83-166 for one scalar path (length,) with NaN = missing."""
    L = x.numel()
    observed = ~torch.isnan(x)
    if not bool(observed.any()):
        z = torch.zeros(L - 1, dtype=x.dtype)
        return z, z.clone(), z.clone(), z.clone()
    x = x.clone()
    obs = observed.nonzero().flatten().tolist()
    if version == 0:                       # impute only the two end points
        if not observed[0]:
            x[0] = x[obs[0]]
        if not observed[-1]:
            x[-1] = x[obs[-1]]
    else:                                  # fill backward / forward from the first / last observation
        x[:obs[0]] = x[obs[0]]
        x[obs[-1] + 1:] = x[obs[-1]]
    keep = ~torch.isnan(x)
    tk, xk = t[keep], x[keep]
    pa, pb, pc, pd = _natural_pieces(tk, xk)
    a, b, c, d = [], [], [], []
    k = -1
    nxt = 0                                # index into tk of the next knot
    for j in range(L - 1):
        if t[j] >= tk[nxt]:
            prev_time = tk[nxt]
            nxt += 1
            k += 1
        offset = prev_time - t[j]
        inner = (0.5 * pc[k] - pd[k] * offset / 3) * offset
        a.append(pa[k] + (inner - pb[k]) * offset)
        b.append(pb[k] + (pd[k] * offset - pc[k]) * offset)
        c.append(pc[k] - 2 * pd[k] * offset)
        d.append(pd[k])
    return torch.stack(a), torch.stack(b), torch.stack(c), torch.stack(d)


def natural_cubic_coeffs(x, t=None, version=1):
    """natural_cubic_coeffs (version 1) / natural_cubic_spline_coeffs (version 0), interpolation_cubic.py:172-266.
    Without missing values every scalar path goes through ``_natural_pieces`` (the vectorised branch of the reference
    performs the same operations per element)."""
    t = check_path(x, t)
    flat = x.transpose(-1, -2).reshape(-1, x.size(-2))
    has_nan = bool(torch.isnan(x).any())
    parts = [(_natural_scalar_path(t, row, version) if has_nan else _natural_pieces(t, row)) for row in flat]
    out = []
    for which in range(4):
        block = torch.stack([p[which] for p in parts]).reshape(*x.shape[:-2], x.size(-1), x.size(-2) - 1)
        out.append(block.transpose(-1, -2))
    return torch.cat(out, dim=-1)


# ----------------------------------------------------------------------------- lookup
def locate(t, knots, n_intervals, dtype, device):
    """interpolation_cubic.py:315-322.  Returns (frac, index[int64]).

    ``bucketize`` with right=False: a query exactly on knot k lands in interval k-1
    with frac == knot spacing; clamped to [0, n_intervals-1], extrapolating outside."""
    t = torch.as_tensor(t, dtype=dtype, device=device)
    index = torch.bucketize(t.detach(), knots.detach()).sub(1).clamp(0, n_intervals - 1)
    frac = t - knots[index]
    return frac, index


def cubic_value(coeffs, knots, t):
    """interpolation_cubic.py:324-329 on the packed coefficient tensor."""
    C = coeffs.size(-1) // 4
    a, b, two_c, three_d = (coeffs[..., :C], coeffs[..., C:2 * C], coeffs[..., 2 * C:3 * C], coeffs[..., 3 * C:])
    frac, index = locate(t, knots, coeffs.size(-2), coeffs.dtype, coeffs.device)
    frac = frac.unsqueeze(-1)
    inner = 0.5 * two_c[..., index, :] + three_d[..., index, :] * frac / 3
    inner = b[..., index, :] + inner * frac
    return a[..., index, :] + inner * frac


def cubic_slope(coeffs, knots, t):
    """interpolation_cubic.py:331-336."""
    C = coeffs.size(-1) // 4
    b, two_c, three_d = (coeffs[..., C:2 * C], coeffs[..., 2 * C:3 * C], coeffs[..., 3 * C:])
    frac, index = locate(t, knots, coeffs.size(-2), coeffs.dtype, coeffs.device)
    frac = frac.unsqueeze(-1)
    inner = two_c[..., index, :] + three_d[..., index, :] * frac
    return b[..., index, :] + inner * frac


def linear_slopes(coeffs, knots):
    """interpolation_linear.py:189 (constructor pre-computation)."""
    return (coeffs[..., 1:, :] - coeffs[..., :-1, :]) / (knots[1:] - knots[:-1]).unsqueeze(-1)


def linear_value(coeffs, knots, t):
    """interpolation_linear.py:212-220."""
    frac, index = locate(t, knots, coeffs.size(-2) - 1, coeffs.dtype, coeffs.device)
    frac = frac.unsqueeze(-1)
    lo = coeffs[..., index, :]
    hi = coeffs[..., index + 1, :]
    width = knots[index + 1] - knots[index]
    return lo + frac * (hi - lo) / width.unsqueeze(-1)


def linear_slope(coeffs, knots, t):
    """interpolation_linear.py:222-225."""
    _, index = locate(t, knots, coeffs.size(-2) - 1, coeffs.dtype, coeffs.device)
    return linear_slopes(coeffs, knots)[..., index, :]


# ----------------------------------------------------------------------------- module shells
class CubicPath(torch.nn.Module):
    """interpolation_cubic.py:282-336 as a thin nn.Module over the functions above."""

    def __init__(self, coeffs, t=None):
        super().__init__()
        if t is None:
            t = torch.linspace(0, coeffs.size(-2), coeffs.size(-2) + 1, dtype=coeffs.dtype, device=coeffs.device)
        if (coeffs.size(-1) // 4) * 4 != coeffs.size(-1):
            raise ValueError("Passed invalid coeffs.")
        self.register_buffer("_t", t)
        self.register_buffer("_coeffs", coeffs)

    @property
    def grid_points(self):
        return self._t

    @property
    def interval(self):
        return torch.stack([self._t[0], self._t[-1]])

    def _interpret_t(self, t):
        return locate(t, self._t, self._coeffs.size(-2), self._coeffs.dtype, self._coeffs.device)

    def evaluate(self, t):
        return cubic_value(self._coeffs, self._t, t)

    def derivative(self, t):
        return cubic_slope(self._coeffs, self._t, t)


class LinearPath(torch.nn.Module):
    """interpolation_linear.py:177-225."""

    def __init__(self, coeffs, t=None):
        super().__init__()
        if t is None:
            t = torch.linspace(0, coeffs.size(-2) - 1, coeffs.size(-2), dtype=coeffs.dtype, device=coeffs.device)
        self.register_buffer("_t", t)
        self.register_buffer("_coeffs", coeffs)
        self.register_buffer("_derivs", linear_slopes(coeffs, t))

    @property
    def grid_points(self):
        return self._t

    @property
    def interval(self):
        return torch.stack([self._t[0], self._t[-1]])

    def _interpret_t(self, t):
        return locate(t, self._t, self._derivs.size(-2), self._derivs.dtype, self._derivs.device)

    def evaluate(self, t):
        return linear_value(self._coeffs, self._t, t)

    def derivative(self, t):
        _, index = self._interpret_t(t)
        return self._derivs[..., index, :]
