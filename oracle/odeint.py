"""Oracle: ODE integrators the reference delegates to (CPU, torch eager).  TEST INFRASTRUCTURE.

The reference never integrates anything itself: ``torchcde/solver.py:226-227`` hands a
``_VectorField`` to ``torchdiffeq.odeint`` / ``odeint_adjoint``.  ``torchdiffeq``
(requirement ``>=0.2.0``, unpinned -- reference ``setup.py:51``) is NOT vendored under
/root/reference and NOT installed here, so this file restates its published algorithm
(torchdiffeq 0.2.x ``_impl/{odeint,misc,solvers,fixed_grid,rk_common,dopri5,interp,adjoint}.py``)
from the description in SURVEY.md Appendix A.

    **PARITY UNPINNED** against the real third-party package.

What anchors it instead (tests/test_oracle.py): 4th-order convergence of ``rk4`` and
tolerance-convergence of ``dopri5`` to closed-form linear-CDE solutions; adjoint gradients
vs. autograd straight through the solver; float64 ``gradcheck``; and the reference's own
``test_cdeint.py`` / ``test_tricks.py`` run with this module registered as ``torchdiffeq``
(``oracle/make_golden.py --run-reference-tests``).

Semantics that matter for bit-level agreement with the HIP kernels:
  * fixed grid: ``n = ceil((t[-1]-t[0])/h + 1)``; ``grid = arange(n)*h + t[0]``; ``grid[-1] = t[-1]``
  * the user function always sees ``t`` cast to the state dtype
  * ``rk4`` is the 3/8 rule with the exact association order written in ``_rk4_38_increment``
  * outputs between grid points: linear interpolation, end points returned verbatim
  * decreasing ``t`` is solved as increasing ``-t`` with the field ``-f(-t, y)``
  * adjoint: augmented state (vjp_t, y, a_y, a_params...) integrated backwards per output
    interval with the forward method/options; ``y`` reset to the stored forward value and
    ``a_y`` bumped by the incoming gradient at every output time.
"""
import bisect

import torch

_THIRD = 1 / 3
_TWO_THIRDS = 2 / 3

_NONE, _PREV, _NEXT = 0, 1, 2


# ----------------------------------------------------------------------------- helpers
def _rms(x):
    return x.abs().pow(2).mean().sqrt()


def _nudge(t, direction):
    # one ulp towards direction (in t's own dtype); gradient passes straight through
    with torch.no_grad():
        moved = torch.nextafter(t, t + direction)
        delta = moved - t                       # exactly one ulp, so t + delta == moved bit-for-bit
    return t + delta


class _Field:
    """Presents the user function the way torchdiffeq's wrappers do: optional tuple
    flattening, optional time reversal, time cast to the state dtype, optional ulp nudge."""

    def __init__(self, func, shapes=None, reverse=False):
        self.func = func
        self.shapes = shapes
        self.reverse = reverse

    def __call__(self, t, y, perturb=_NONE):
        t = t.to(y.dtype)
        if perturb == _NEXT:
            t = _nudge(t, 1)
        elif perturb == _PREV:
            t = _nudge(t, -1)
        if self.reverse:
            t = -t
        if self.shapes is None:
            out = self.func(t, y)
        else:
            out = self.func(t, _unflatten(y, (), self.shapes))
            out = torch.cat([o.reshape(-1) for o in out])
        if self.reverse:
            out = -1.0 * out
        return out


def _unflatten(flat, lead, shapes):
    pieces, offset = [], 0
    for shape in shapes:
        n = 1
        for s in shape:
            n *= s
        pieces.append(flat[..., offset:offset + n].reshape(tuple(lead) + tuple(shape)))
        offset += n
    return tuple(pieces)


# ----------------------------------------------------------------------------- fixed grid
def _grid_from_step(t, step_size):
    start, end = t[0], t[-1]
    n = torch.ceil((end - start) / step_size + 1).item()
    grid = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + start
    grid[-1] = t[-1]
    return grid


def _rk4_38_increment(f, t0, dt, t1, y0, perturb):
    k1 = f(t0, y0, perturb=_NEXT if perturb else _NONE)
    k2 = f(t0 + dt * _THIRD, y0 + dt * k1 * _THIRD)
    k3 = f(t0 + dt * _TWO_THIRDS, y0 + dt * (k2 - k1 * _THIRD))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3), perturb=_PREV if perturb else _NONE)
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


def _midpoint_increment(f, t0, dt, t1, y0, perturb):
    half = 0.5 * dt
    k1 = f(t0, y0, perturb=_NEXT if perturb else _NONE)
    return dt * f(t0 + half, y0 + k1 * half)


def _euler_increment(f, t0, dt, t1, y0, perturb):
    return dt * f(t0, y0, perturb=_NEXT if perturb else _NONE)


_FIXED = {"rk4": _rk4_38_increment, "midpoint": _midpoint_increment, "euler": _euler_increment}


def _integrate_fixed(increment, f, y0, t, step_size=None, grid_constructor=None, perturb=False, interp="linear"):
    if interp != "linear":
        raise NotImplementedError("oracle: only linear output interpolation is restated")
    if step_size is not None and grid_constructor is not None:
        raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")
    if step_size is not None:
        grid = _grid_from_step(t, step_size)
    elif grid_constructor is not None:
        grid = grid_constructor(f, y0, t)
    else:
        grid = t
    assert grid[0] == t[0] and grid[-1] == t[-1]
    out = torch.empty(len(t), *y0.shape, dtype=y0.dtype, device=y0.device)
    out[0] = y0
    j = 1
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dt = t1 - t0
        y1 = y0 + increment(f, t0, dt, t1, y0, perturb)
        while j < len(t) and t1 >= t[j]:
            tj = t[j]
            if tj == t0:
                out[j] = y0
            elif tj == t1:
                out[j] = y1
            else:
                slope = (tj - t0) / (t1 - t0)
                out[j] = y0 + slope * (y1 - y0)
            j += 1
        y0 = y1
    return out


# ----------------------------------------------------------------------------- dopri5
_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.]
_DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP_C_ERR = [
    35 / 384 - 1951 / 21600,
    0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1. / 60.,
]
_DP_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2,
]


def _sorted_after(values, t0):
    values = values[values >= t0]
    return torch.sort(values).values


class _Dopri5:
    order = 5

    def __init__(self, f, y0, rtol, atol, norm, first_step=None, step_t=None, jump_t=None, safety=0.9, ifactor=10.0,
                 dfactor=0.2, max_num_steps=2 ** 31 - 1, min_step=0, max_step=float("inf"), dtype=torch.float64,
                 replay_steps=None, replay_attempts=None, attempt_probe=None):
        tdtype = torch.promote_types(dtype, y0.dtype)
        dev = y0.device
        self.f, self.y0, self.norm, self.tdtype = f, y0, norm, tdtype
        as_t = lambda v: torch.as_tensor(v, dtype=tdtype, device=dev)
        self.rtol, self.atol = as_t(rtol), as_t(atol)
        self.min_step, self.max_step = as_t(min_step), as_t(max_step)
        self.first_step = None if first_step is None else as_t(first_step)
        self.safety, self.ifactor, self.dfactor = as_t(safety), as_t(ifactor), as_t(dfactor)
        self.max_num_steps = max_num_steps
        self.step_t = None if step_t is None else as_t(step_t)
        self.jump_t = None if jump_t is None else as_t(jump_t)
        yd = dict(dtype=y0.dtype, device=dev)
        self.alpha = torch.tensor(_DP_ALPHA, **yd)
        self.beta = [torch.tensor(b, **yd) for b in _DP_BETA]
        self.c_sol = torch.tensor(_DP_C_SOL, **yd)
        self.c_err = torch.tensor(_DP_C_ERR, **yd)
        self.c_mid = torch.tensor(_DP_C_MID, **yd)
        self.n_accept = 0
        self.n_reject = 0
        self.accepted = []           # (t0, t1, on_jump) of every accepted step (test infrastructure: the GPU trace's twin)
        # TEST INFRASTRUCTURE, not a torchdiffeq option: a list / (n, 3) tensor of accepted (t0, t1, on_jump) steps.  The
        # controller is bypassed and exactly these steps are taken -- used to check a SAMPLE of a large batch against
        # the step sequence the batch-global controller chose for the WHOLE batch (tests/test_gpu_01_atsize.py, config 4).
        self.replay_steps = None if replay_steps is None else torch.as_tensor(replay_steps, dtype=tdtype).reshape(-1, 3)
        # TEST INFRASTRUCTURE as well: EVERY attempt another solver made -- rows (t0, t1, on_jump, accepted, ...), rejected
        # ones included -- is re-made from the state THIS solver has at t0; the error ratio this solver computes for it is
        # recorded in `self.ratios` (and the initial step it would have chosen in `self.first_dt`), the attempt's own
        # accept flag decides whether the state moves.  A list is consumed one entry per solver (the adjoint pass creates
        # one solver per output interval, last interval first).
        if isinstance(replay_attempts, list):
            replay_attempts = replay_attempts.pop(0)
        self.replay_attempts = None if replay_attempts is None else torch.as_tensor(replay_attempts, dtype=tdtype)
        self.ratios, self.first_dt = [], None
        # TEST INFRASTRUCTURE: called as attempt_probe(y0, y1, y1_err) (flat states) for every re-made attempt.  A batch-global
        # error norm cannot be formed on a CHUNK of a large batch; the probe lets a test collect each chunk's share of the
        # norm's sums and assemble the whole batch's error ratio (tests/test_gpu_01_atsize.py).
        self.attempt_probe = attempt_probe

    # -- initial step (Hairer), order argument = self.order - 1
    def _initial_step(self, t0, f0):
        y0, rtol, atol, norm = self.y0, self.rtol, self.atol, self.norm
        dtype, dev, tdtype = y0.dtype, y0.device, t0.dtype
        scale = atol + torch.abs(y0) * rtol
        d0 = norm(y0 / scale).abs()
        d1 = norm(f0 / scale).abs()
        if d0 < 1e-5 or d1 < 1e-5:
            h0 = torch.tensor(1e-6, dtype=dtype, device=dev)
        else:
            h0 = 0.01 * d0 / d1
        h0 = h0.abs()
        y1 = y0 + h0 * f0
        f1 = self.f(t0 + h0, y1)
        d2 = torch.abs(norm((f1 - f0) / scale) / h0)
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = torch.max(torch.tensor(1e-6, dtype=dtype, device=dev), h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1, d2)) ** (1. / float(self.order))
        h1 = h1.abs()
        return torch.min(100 * h0, h1).to(tdtype)

    def _rk_step(self, y0, f0, t0, dt, t1):
        t0, dt, t1 = t0.to(y0.dtype), dt.to(y0.dtype), t1.to(y0.dtype)
        # torchdiffeq fills one (..., 7) buffer in place; stacking a list is the autograd-safe equivalent
        ks = [f0]
        yi = y0
        for i, (alpha_i, beta_i) in enumerate(zip(self.alpha, self.beta)):
            if alpha_i == 1.:
                ti, perturb = t1, _PREV
            else:
                ti, perturb = t0 + alpha_i * dt, _NONE
            yi = y0 + torch.stack(ks, dim=-1).matmul(beta_i * dt).view_as(f0)
            ks.append(self.f(ti, yi, perturb=perturb))
        k = torch.stack(ks, dim=-1)
        y1 = yi                      # FSAL property of Dormand-Prince: c_sol == beta[-1]
        f1 = ks[-1]
        y1_err = k.matmul(dt * self.c_err)
        return y1, f1, y1_err, k

    def _fit_dense(self, y0, y1, k, dt):
        dt = dt.type_as(y0)
        y_mid = y0 + k.matmul(dt * self.c_mid).view_as(y0)
        f0, f1 = k[..., 0], k[..., -1]
        a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
        b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
        c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
        d = dt * f0
        e = y0
        return [e, d, c, b, a]

    @staticmethod
    def _eval_dense(coeffs, t0, t1, t):
        x = ((t - t0) / (t1 - t0)).to(coeffs[0].dtype)
        total = coeffs[0] + x * coeffs[1]
        xp = x
        for c in coeffs[2:]:
            xp = xp * x
            total = total + xp * c
        return total

    def _next_dt(self, last, ratio):
        if ratio == 0:
            return last * self.ifactor
        dfactor = self.dfactor
        if ratio < 1:
            dfactor = torch.ones((), dtype=last.dtype, device=last.device)
        ratio = ratio.type_as(last)
        exponent = torch.tensor(self.order, dtype=last.dtype, device=last.device).reciprocal()
        factor = torch.min(self.ifactor, torch.max(self.safety / ratio ** exponent, dfactor))
        return last * factor

    def _integrate_replay(self, t):
        y0 = self.y0
        out = torch.empty(len(t), *y0.shape, dtype=y0.dtype, device=y0.device)
        out[0] = y0
        t = t.to(self.tdtype)
        y, f = y0, self.f(t[0], y0)
        i = 1
        for t0, t1, on_jump in self.replay_steps:
            if i >= len(t):
                break
            y1, f1, _, k = self._rk_step(y, f, t0, t1 - t0, t1)
            dense = self._fit_dense(y, y1, k, t1 - t0)
            if on_jump != 0:                            # the step was clipped onto a jump: f re-evaluated just after it
                f1 = self.f(t1, y1, perturb=_NEXT)
            self.n_accept += 1
            self.accepted.append((t0.item(), t1.item(), float(on_jump)))
            while i < len(t) and not (t[i] > t1):
                out[i] = self._eval_dense(dense, t0, t1, t[i])
                i += 1
            y, f = y1, f1
        assert i == len(t), "replayed steps end before the last output time"
        return out

    def _integrate_attempts(self, t):
        y0 = self.y0
        out = torch.empty(len(t), *y0.shape, dtype=y0.dtype, device=y0.device)
        out[0] = y0
        t = t.to(self.tdtype)
        y, f = y0, self.f(t[0], y0)
        self.first_dt = self._initial_step(t[0], f)
        i = 1
        for row in self.replay_attempts:
            t0, t1, on_jump, accepted = row[0], row[1], row[2], row[3]
            dt = t1 - t0
            y1, f1, y1_err, k = self._rk_step(y, f, t0, dt, t1)
            tol = self.atol + self.rtol * torch.max(y.abs(), y1.abs())
            self.ratios.append(float(self.norm(y1_err / tol).abs()))
            if self.attempt_probe is not None:
                self.attempt_probe(y, y1, y1_err)
            if accepted != 0:
                dense = self._fit_dense(y, y1, k, dt)
                if on_jump != 0:
                    f1 = self.f(t1, y1, perturb=_NEXT)
                self.n_accept += 1
                self.accepted.append((t0.item(), t1.item(), float(on_jump)))
                while i < len(t) and not (t[i] > t1):
                    out[i] = self._eval_dense(dense, t0, t1, t[i])
                    i += 1
                y, f = y1, f1
            else:
                self.n_reject += 1
        assert i == len(t), "replayed attempts end before the last output time"
        return out

    def integrate(self, t):
        if self.replay_attempts is not None:
            return self._integrate_attempts(t)
        if self.replay_steps is not None:
            return self._integrate_replay(t)
        y0 = self.y0
        out = torch.empty(len(t), *y0.shape, dtype=y0.dtype, device=y0.device)
        out[0] = y0
        t = t.to(self.tdtype)
        f0 = self.f(t[0], y0)
        dt = self._initial_step(t[0], f0) if self.first_step is None else self.first_step
        empty = torch.tensor([], dtype=self.tdtype, device=y0.device)
        step_t = empty if self.step_t is None else _sorted_after(self.step_t, t[0]).to(self.tdtype)
        jump_t = empty if self.jump_t is None else _sorted_after(self.jump_t, t[0]).to(self.tdtype)
        if (torch.cat([step_t, jump_t]).unique(return_counts=True)[1] > 1).any():
            raise ValueError("`step_t` and `jump_t` must not have any repeated elements between them.")
        i_step = min(bisect.bisect(step_t.tolist(), t[0]), len(step_t) - 1)
        i_jump = min(bisect.bisect(jump_t.tolist(), t[0]), len(jump_t) - 1)
        # state: (y at t_hi, f at t_hi, t_lo, t_hi, dt, dense coefficients over [t_lo, t_hi])
        y, f, t_lo, t_hi, dense = y0, f0, t[0], t[0], [y0] * 5
        for i in range(1, len(t)):
            target = t[i]
            n = 0
            while target > t_hi:
                assert n < self.max_num_steps, "max_num_steps exceeded"
                n += 1
                # ---- one attempted step from t_hi
                t0 = t_hi
                if not torch.isfinite(dt):
                    dt = self.min_step
                dt = dt.clamp(self.min_step, self.max_step)
                t1 = t0 + dt
                assert t0 + dt > t0, "underflow in dt {}".format(dt.item())
                assert torch.isfinite(y).all(), "non-finite values in state `y`"
                on_step = False
                if len(step_t):
                    nxt = step_t[i_step]
                    on_step = bool(t0 < nxt < t0 + dt)
                    if on_step:
                        t1 = nxt
                        dt = t1 - t0
                on_jump = False
                if len(jump_t):
                    nxt = jump_t[i_jump]
                    on_jump = bool(t0 < nxt < t0 + dt)
                    if on_jump:
                        on_step = False
                        t1 = nxt
                        dt = t1 - t0
                y1, f1, y1_err, k = self._rk_step(y, f, t0, dt, t1)
                tol = self.atol + self.rtol * torch.max(y.abs(), y1.abs())
                ratio = self.norm(y1_err / tol).abs()
                accept = bool(ratio <= 1)
                if dt > self.max_step:
                    accept = False
                if dt <= self.min_step:
                    accept = True
                if accept:
                    self.n_accept += 1
                    self.accepted.append((t0.item(), t1.item(), float(on_jump)))
                    dense = self._fit_dense(y, y1, k, dt)
                    if on_step and i_step != len(step_t) - 1:
                        i_step += 1
                    if on_jump:
                        if i_jump != len(jump_t) - 1:
                            i_jump += 1
                        f1 = self.f(t1, y1, perturb=_NEXT)
                    y, f, t_lo, t_hi = y1, f1, t0, t1
                else:
                    self.n_reject += 1
                    t_lo, t_hi = t0, t0
                dt = self._next_dt(dt, ratio).clamp(self.min_step, self.max_step)
            out[i] = self._eval_dense(dense, t_lo, t_hi, target)
        return out


# ----------------------------------------------------------------------------- front end
def _prepare(func, y0, t, options):
    shapes = None
    if not isinstance(y0, torch.Tensor):
        shapes = [tuple(p.shape) for p in y0]
        y0 = torch.cat([p.reshape(-1) for p in y0])
    if not torch.is_floating_point(y0):
        raise TypeError("`y0` must be a floating point Tensor")
    if not (isinstance(t, torch.Tensor) and t.dim() == 1):
        raise ValueError("t must be one dimensional")
    if t.device != y0.device:
        t = t.to(y0.device)
    reverse = len(t) > 1 and bool(t[0] > t[1])
    if reverse:
        t = -t
        for key in ("step_t", "jump_t"):
            if options.get(key) is not None:
                options[key] = -torch.as_tensor(options[key]).flip(0)
    if not (t[1:] > t[:-1]).all():
        raise ValueError("t must be strictly increasing or decreasing")
    return _Field(func, shapes, reverse), y0, t, shapes


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None):
    """Restatement of ``torchdiffeq.odeint`` -> tensor (len(t), *y0.shape) or tuple of them."""
    if event_fn is not None:
        raise NotImplementedError("oracle: event handling is outside the hot path")
    options = {} if options is None else dict(options)
    method = "dopri5" if method is None else method
    f, y0_flat, t_run, shapes = _prepare(func, y0, t, options)
    norm = options.pop("norm", None)
    if norm is None:
        if shapes is None:
            norm = _rms
        else:
            norm = lambda flat: max(_rms(p) for p in _unflatten(flat, (), shapes))
    elif shapes is not None:
        user_norm = norm
        norm = lambda flat: user_norm(_unflatten(flat, (), shapes))
    if method in _FIXED:
        sol = _integrate_fixed(_FIXED[method], f, y0_flat, t_run, **options)
    elif method == "dopri5":
        sol = _Dopri5(f, y0_flat, rtol, atol, norm, **options).integrate(t_run)
    else:
        raise ValueError("oracle: method {!r} is not restated".format(method))
    if shapes is not None:
        sol = _unflatten(sol, (len(t),), shapes)
    return sol


class _Adjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, y0, t, *adjoint_params):
        ctx.cfg = cfg
        with torch.no_grad():
            ans = odeint(cfg["func"], y0, t, rtol=cfg["rtol"], atol=cfg["atol"], method=cfg["method"],
                         options=cfg["options"])
        ctx.save_for_backward(t, ans, *adjoint_params)
        return ans

    @staticmethod
    def backward(ctx, grad_y):
        cfg = ctx.cfg
        func = cfg["func"]
        t_requires_grad = cfg["t_requires_grad"]
        t, y, *adjoint_params = ctx.saved_tensors
        adjoint_params = tuple(adjoint_params)
        with torch.no_grad():
            aug = [torch.zeros((), dtype=y.dtype, device=y.device), y[-1], grad_y[-1]]
            aug.extend(torch.zeros_like(p) for p in adjoint_params)

            def augmented(time, state):
                yy, aa = state[1], state[2]
                with torch.enable_grad():
                    t_ = time.detach()
                    tt = t_.requires_grad_(True)
                    yy = yy.detach().requires_grad_(True)
                    fe = func(tt if t_requires_grad else t_, yy)
                    vt, vy, *vp = torch.autograd.grad(fe, (tt, yy) + adjoint_params, -aa, allow_unused=True,
                                                      retain_graph=True)
                vt = torch.zeros_like(tt) if vt is None else vt
                vy = torch.zeros_like(yy) if vy is None else vy
                vp = [torch.zeros_like(p) if g is None else g for p, g in zip(adjoint_params, vp)]
                return (vt, fe, vy, *vp)

            time_vjps = torch.empty(len(t), dtype=t.dtype, device=t.device) if t_requires_grad else None
            for i in range(len(t) - 1, 0, -1):
                if t_requires_grad:
                    fe = func(t[i], y[i])
                    dldt = fe.reshape(-1).dot(grad_y[i].reshape(-1))
                    aug[0] -= dldt
                    time_vjps[i] = dldt
                sol = odeint(augmented, tuple(aug), t[i - 1:i + 1].flip(0), rtol=cfg["adjoint_rtol"],
                             atol=cfg["adjoint_atol"], method=cfg["adjoint_method"], options=cfg["adjoint_options"])
                aug = [s[1] for s in sol]
                aug[1] = y[i - 1]
                aug[2] += grad_y[i - 1]
            if t_requires_grad:
                time_vjps[0] = aug[0]
        return (None, aug[2], time_vjps, *aug[3:])


def odeint_adjoint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None,
                   adjoint_rtol=None, adjoint_atol=None, adjoint_method=None, adjoint_options=None,
                   adjoint_params=None):
    """Restatement of ``torchdiffeq.odeint_adjoint`` (continuous adjoint)."""
    if event_fn is not None:
        raise NotImplementedError("oracle: event handling is outside the hot path")
    if adjoint_params is None and not isinstance(func, torch.nn.Module):
        raise ValueError("func must be an instance of nn.Module to specify the adjoint parameters; alternatively "
                         "they can be specified explicitly via the `adjoint_params` argument. If there are no "
                         "parameters then it is allowable to set `adjoint_params=()`.")
    adjoint_rtol = rtol if adjoint_rtol is None else adjoint_rtol
    adjoint_atol = atol if adjoint_atol is None else adjoint_atol
    adjoint_method = method if adjoint_method is None else adjoint_method
    if adjoint_options is None:
        adjoint_options = {k: v for k, v in options.items() if k != "norm"} if options is not None else {}
    else:
        adjoint_options = dict(adjoint_options)
    if adjoint_params is None:
        adjoint_params = tuple(func.parameters())
    else:
        adjoint_params = tuple(adjoint_params)
    adjoint_params = tuple(p for p in adjoint_params if p.requires_grad)

    shapes = None
    if not isinstance(y0, torch.Tensor):
        shapes = [tuple(p.shape) for p in y0]
        y0_in = torch.cat([p.reshape(-1) for p in y0])
        user = func
        func_flat = lambda tt, flat: torch.cat([o.reshape(-1) for o in user(tt, _unflatten(flat, (), shapes))])
    else:
        y0_in, func_flat = y0, func
    # torchdiffeq's wrapper hands the user function a time already cast to the state dtype
    func_uncast = func_flat
    func_flat = lambda tt, yy: func_uncast(tt.to(yy.dtype), yy)

    # adjoint_options=dict(norm="seminorm"): torchdiffeq replaces the string by max(|vjp_t|, rms(y), rms(a_y)) -- the
    # parameter blocks do not take part in the step-size control
    if adjoint_options.get("norm") == "seminorm":
        adjoint_options["norm"] = lambda parts: max(parts[0].abs(), _rms(parts[1]), _rms(parts[2]))
    # default adjoint norm for adaptive methods: max(|vjp_t|, rms(y), rms(a_y), max_p rms(a_p))
    if "norm" not in adjoint_options:
        def adjoint_norm(parts):
            tt, yy, aa, *pp = parts
            extra = max([_rms(p) for p in pp]) if pp else 0.0
            return max(tt.abs(), _rms(yy), _rms(aa), extra)
        adjoint_options["norm"] = adjoint_norm

    cfg = dict(func=func_flat, rtol=rtol, atol=atol, method=method, options=options, adjoint_rtol=adjoint_rtol,
               adjoint_atol=adjoint_atol, adjoint_method=adjoint_method, adjoint_options=adjoint_options,
               t_requires_grad=t.requires_grad)
    ans = _Adjoint.apply(cfg, y0_in, t, *adjoint_params)
    if shapes is not None:
        ans = _unflatten(ans, (len(t),), shapes)
    return ans
