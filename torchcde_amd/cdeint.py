"""``cdeint``: the drop-in solver front-end (host side of K2 / K3).

Mirrors reference ``torchcde/solver.py:144-245`` -- same signature, defaults, ValueError messages
and output layout ``(..., len(t), hidden)`` -- and replaces what lies behind it
(``_VectorField.forward`` solver.py:117-135 and ``torchdiffeq.odeint[_adjoint]`` solver.py:226-227)
by ONE fused HIP kernel per direction.  What the native path covers is stated explicitly; anything
else raises ``NotImplementedError`` -- it never falls back to an eager/CPU computation.

Native scope (round 1):
  * X: ``torchcde_amd.CubicSpline`` or ``torchcde_amd.LinearInterpolation``
  * func: the affine family recognised by ``torchcde_amd.fields`` (Linear(H, H*C) [+ tanh] viewed (..., H, C))
  * backend "torchdiffeq", ``method='rk4'`` (torchdiffeq's 3/8-rule), ``options={'step_size': h}`` or no options
    (grid = t, torchdiffeq's behaviour), increasing ``t``, tensor state; gradients by the continuous adjoint
    (``adjoint=True``) for z0 and the field's weight / bias
  * ``method='dopri5'`` (also the default when no method is passed, as in the reference): adaptive solve with
    torchdiffeq's batch-global controller, ``rtol/atol`` and ``options={'jump_t': ...}``; fused forward
  * everything else -- arbitrary ``func`` modules (MLPs ...), ``midpoint``/``euler``, ``adjoint=False`` backprop,
    gradients through ``dopri5`` -- runs step-wise (``torchcde_amd/stepwise.py``): host-driven stepping on the GPU
    with the native control-derivative and contraction kernels under every vector-field evaluation
"""
import ctypes
import sys
import threading
import types
import warnings
import weakref

import torch
from torch.autograd.function import once_differentiable as _once_differentiable

from . import _lib, dispatch
from .fields import probe
from .paths import _NativePath

_GRAD_WARNING = ("One of the inputs to the control path X requires gradients but "
                 "`kwargs['adjoint_params']` has not been passed. This is probably a mistake: these "
                 "inputs will not receive a gradient when using the adjoint method. Either have the input "
                 "not require gradients (if that was unintended), or include it (and every other "
                 "parameter needing gradients) in `adjoint_params`. For example:\n"
                 "```\n"
                 "coeffs = ...\n"
                 "func = ...\n"
                 "X = CubicSpline(coeffs)\n"
                 "adjoint_params = tuple(func.parameters()) + (coeffs,)\n"
                 "cdeint(X=X, func=func, ..., adjoint_params=adjoint_params)\n"
                 "```")


# ------------------------------------------------------------------------------------------ per-thread call state
class _CallState:
    """What a caller may ask about / configure for ITS OWN cdeint calls: the step statistics of its most recent adaptive
    solve and fused adaptive backward, whether the step traces are fetched as well (tests), and the list bench.py hands
    in to receive HIP events around the K2 / K3 C-ABI calls.  One per calling thread (like dispatch.last()); a plan
    remembers the state of the thread that built it, because autograd runs its backward on an engine thread."""
    __slots__ = ("dopri5", "dopri5_adjoint", "record", "event_log")

    def __init__(self):
        self.dopri5, self.dopri5_adjoint, self.record, self.event_log = {}, {}, False, None


_tls = threading.local()


def _state():
    st = getattr(_tls, "state", None)
    if st is None:
        st = _tls.state = _CallState()
    return st


# ------------------------------------------------------------------------------------------ time grids
def _fixed_grid(t, step_size):
    """torchdiffeq's fixed-grid constructor, evaluated with the same torch ops on
    the host: ``n = ceil((t[-1]-t[0])/h + 1)``; ``grid = arange(n)*h + t[0]``; ``grid[-1] = t[-1]``."""
    if step_size is None:
        grid = t.clone()
    else:
        n = torch.ceil((t[-1] - t[0]) / step_size + 1).item()
        grid = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + t[0]
        grid[-1] = t[-1]
    if not (grid[0] == t[0] and grid[-1] == t[-1]):
        raise AssertionError("time grid does not span [t[0], t[-1]]")
    return grid


def _parse_fixed_options(options, what):
    options = {} if options is None else dict(options)
    step_size = options.pop("step_size", None)
    if options.pop("grid_constructor", None) is not None:
        raise NotImplementedError("torchcde_amd: %s option 'grid_constructor' is not supported natively" % what)
    if options.pop("perturb", False):
        raise NotImplementedError("torchcde_amd: %s option 'perturb' is not supported natively" % what)
    if options.pop("interp", "linear") != "linear":
        raise NotImplementedError("torchcde_amd: only interp='linear' is supported natively")
    options.pop("norm", None)  # only used by adaptive solvers
    if options:
        raise NotImplementedError("torchcde_amd: unsupported %s options %s" % (what, sorted(options)))
    return step_size


def _strip_noops(opts, fixed):
    """Options as the fused paths see them: keys whose value is None are absent (torchdiffeq's `options.get(key)`
    semantics), and for the fixed-grid methods so are torchdiffeq's defaults spelled out -- interp='linear',
    perturb=False -- and `norm`, which only adaptive solvers read.  ADVICE round 3: the dispatch test and the plans used
    to judge these differently (a fused call went step-wise over interp='linear'; safety=None raised a TypeError)."""
    if not isinstance(opts, dict):
        return opts
    out = {k: v for k, v in opts.items() if v is not None}
    if fixed:
        if out.get("interp") == "linear":
            del out["interp"]
        if out.get("perturb") is False:
            del out["perturb"]
        out.pop("norm", None)
    return out


_host_copies = {}      # id(time tensor) -> (weakref, version, host copy): avoids a D2H sync per call
_grid_cache = {}       # (times, dtype, steps, device) -> device-resident solver grids


def _to_host(t):
    """CPU copy of a (small) time tensor.  A GPU tensor costs one synchronising D2H copy the first time it is
    seen; ``CubicSpline.interval`` hands out one cached tensor object, so steady-state calls do not sync.
    (Keyed by id + weakref: tensors cannot be WeakKeyDictionary keys because ``==`` is elementwise.)"""
    if not t.is_cuda:
        return t.detach()
    key = id(t)
    hit = _host_copies.get(key)
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    host = t.detach().cpu()
    if len(_host_copies) > 256:
        for k in [k for k, v in _host_copies.items() if v[0]() is None]:
            del _host_copies[k]
    _host_copies[key] = (weakref.ref(t), t._version, host)
    return host


class _Grids:
    """Solver grids for one (output times, step sizes) combination, resident on the device."""

    def __init__(self, t_host, step_size, adjoint_step_size, device):
        self.n_out = t_host.numel()
        self.time_dtype = t_host.dtype
        self.t_out = t_host.to(device)
        self.grid = _fixed_grid(t_host, step_size).to(device)
        # reversed-time grids, one per output interval, in processing order i = T-1 .. 1
        pieces, offsets = [], [0]
        for i in range(self.n_out - 1, 0, -1):
            seg_t = -(t_host[i - 1:i + 1].flip(0))     # torchdiffeq: t[i-1:i+1].flip(0) is decreasing -> solved on -t
            pieces.append(_fixed_grid(seg_t, adjoint_step_size))
            offsets.append(offsets[-1] + pieces[-1].numel())
        sgrid = torch.cat(pieces) if pieces else torch.zeros(0, dtype=t_host.dtype)
        self.n_sgrid = sgrid.numel()
        self.sgrid = sgrid.to(device)
        self.seg_off = torch.tensor(offsets, dtype=torch.int64).to(device)
        self.seg_off_host = offsets
        self.seg_off_c = (ctypes.c_int64 * len(offsets))(*offsets)      # host copy handed to the C ABI (wide shapes)
        self._t_host, self._step_size, self._device = t_host, step_size, device
        self._backprop = None

    def backprop_lists(self):
        """What the discrete backward (adjoint=False, cde_rk4_backprop_linear) needs besides the stored stages: the step
        sizes as the forward kernel rounds them, and per grid node the (output index, weight) pairs of torchdiffeq's linear
        output interpolation transposed (fixed-grid solvers: out = y0 + (t - t0) / (t1 - t0) * (y1 - y0), end points
        returned as they are) -- the same walk over (grid, t_out) the forward kernel makes."""
        if self._backprop is None:
            grid, t_out = _fixed_grid(self._t_host, self._step_size), self._t_host
            n_steps = grid.numel() - 1
            nodes = [[] for _ in range(n_steps + 1)]
            nodes[0].append((0, 1.0))
            jout = 1
            for k in range(n_steps):
                t0, t1 = grid[k], grid[k + 1]
                while jout < self.n_out and bool(t1 >= t_out[jout]):
                    tj = t_out[jout]
                    if bool(tj == t0):
                        nodes[k].append((jout, 1.0))
                    elif bool(tj == t1):
                        nodes[k + 1].append((jout, 1.0))
                    else:
                        slope = ((tj - t0) / (t1 - t0)).to(torch.float32)
                        nodes[k].append((jout, float(torch.ones((), dtype=torch.float32) - slope)))
                        nodes[k + 1].append((jout, float(slope)))
                    jout += 1
            ptr, outs, weights = [0], [], []
            for entries in nodes:
                outs += [j for j, _ in entries]
                weights += [w for _, w in entries]
                ptr.append(len(outs))
            dev = self._device
            self.backprop_nodes = nodes                    # host copy: the two-layer sweep runs in chunks that end on these nodes
            self._backprop = (
                (grid[1:] - grid[:-1]).to(torch.float32).to(dev) if n_steps > 0 else torch.zeros(1, dtype=torch.float32, device=dev),
                torch.tensor(ptr, dtype=torch.int64).to(dev), torch.tensor(outs, dtype=torch.int64).to(dev),
                torch.tensor(weights, dtype=torch.float32).to(dev), n_steps)
        return self._backprop


def _grids_for(t_host, step_size, adjoint_step_size, device):
    key = (t_host.numpy().tobytes(), str(t_host.dtype), step_size, adjoint_step_size, str(device))
    hit = _grid_cache.get(key)
    if hit is None:
        if len(_grid_cache) > 64:
            _grid_cache.clear()
        hit = _grid_cache[key] = _Grids(t_host, step_size, adjoint_step_size, device)
    return hit


def _plan_time_gradients(plan, z_saved, grad_out, weight, bias, grad_x, t, want_t, want_knots):
    """Gradients with respect to times from what the adjoint sweep already produced (torchdiffeq's odeint_adjoint,
    SURVEY appendix A.4, for f(t, z) = F(z) dX/dt(t)):
      dL/dt_i (i >= 1) = f(t_i, z_i) . dL/dz_i                       -- one field evaluation per output time
      dL/dt_0          = - sum_i dL/dt_i + int a^T F(z) d2X/dt2 dt    -- the integral in the solver's own quadrature
      dL/d knot_j      = - int_{interval j} a^T F(z) d2X/dt2 dt       -- the reference's `frac = t - t_j` chain
    With a cubic control d2X/dt2 = 2c + 2 (3d) frac, so both integrals are per-interval contractions of the control
    gradient the sweep accumulates anyway: sum_c (2c . dL/db + 2 (3d) . dL/d(2c)).  A piecewise-linear control has no
    such term; its slopes (x_{j+1} - x_j) / h_j depend on the knot times through the widths h_j instead
    (interpolation_linear.py:189): with G_j = dL/d(slope_j), dL/dh_j = -G_j . slope_j / h_j, and G_j / h_j is minus the
    running sum of the knot-value gradient the sweep produced (dL/dx_j = G_{j-1} / h_{j-1} - G_j / h_j telescopes)."""
    B, H, C = plan.B, plan.H, plan.C
    dev = plan.device
    coeffs = plan.coeffs                                              # (B, rows, width)
    if plan.degree == _lib.PATH_CUBIC:
        two_c, three_d = coeffs[..., 2 * C:3 * C], coeffs[..., 3 * C:]
        q = (two_c * grad_x[..., C:2 * C] + 2 * three_d * grad_x[..., 2 * C:3 * C]).sum(-1)      # (B, intervals)
        per_interval = q.sum(0)
    else:
        per_interval = torch.zeros(coeffs.size(1) - 1, dtype=coeffs.dtype, device=dev)
    grad_t = grad_knots = None
    if want_t:
        zs = z_saved.reshape(B, plan.n_out, H)
        go = grad_out.reshape(B, plan.n_out, H)
        vals = [None] * plan.n_out
        total = torch.zeros((), dtype=coeffs.dtype, device=dev)
        for i in range(1, plan.n_out):
            y = torch.nn.functional.linear(zs[:, i], weight, bias)
            if plan.act == _lib.ACT_TANH:
                y = y.tanh()
            dX = plan.path.derivative(plan.t_out[i]).reshape(B, C)
            f = (y.view(B, H, C) * dX.unsqueeze(1)).sum(-1)
            vals[i] = (f * go[:, i]).sum()
            total = total + vals[i]
        vals[0] = per_interval.sum() - total
        # on `t`'s own device: cdeint accepts a CPU `t` next to GPU data (as the reference does)
        grad_t = torch.stack(vals).to(device=t.device, dtype=t.dtype) if plan.n_out > 1 else torch.zeros_like(t)
    if want_knots and plan.degree == _lib.PATH_CUBIC:
        grad_knots = torch.cat([-per_interval, per_interval.new_zeros(1)])
    elif want_knots:
        knots = plan.knots.double()
        values = coeffs.double()
        slopes = (values[:, 1:] - values[:, :-1]) / (knots[1:] - knots[:-1]).unsqueeze(-1)
        dh = (grad_x.double().cumsum(1)[:, :-1] * slopes).sum((0, 2))            # dL/dh_j
        zero = dh.new_zeros(1)
        grad_knots = (torch.cat([zero, dh]) - torch.cat([dh, zero])).to(coeffs.dtype)
    return grad_t, grad_knots


class _Plan:
    """Everything one cdeint call needs besides the differentiable tensors."""

    # bench.py sets `torchcde_amd.cdeint.event_log` (of its thread) to a list to receive ("forward" | "adjoint",
    # start_event, end_event) around the C-ABI calls, recorded on the stream the kernels are launched on.
    def _mark(self):
        if self.owner.event_log is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def __init__(self, path, field, batch, H, C, t, step_size, adjoint_step_size, adjoint, variant, method=_lib.METHOD_RK4):
        self.owner = _state()
        self.method = method             # _lib.METHOD_*: rk4 (3/8 rule), or midpoint / euler on the same kernels
        coeffs, knots, _ = path._native_inputs()
        self.coeffs, self.knots = coeffs, knots
        self.path = path
        self.n_intervals = path._n_intervals()
        self.degree = path._degree
        self.act = field.act
        self.batch, self.B, self.H, self.C = batch, coeffs.size(0), H, C
        self.dtype = coeffs.dtype
        self.device = coeffs.device
        self.adjoint = adjoint
        self.variant = variant
        grids = _grids_for(_to_host(t), step_size, adjoint_step_size, self.device)
        self.grids = grids
        self.time_dtype, self.n_out = grids.time_dtype, grids.n_out
        self.t_out, self.grid = grids.t_out, grids.grid
        self.stage_index = None
        self.stage_frac = None

    time_gradients = _plan_time_gradients

    # forward: K2
    def run_forward(self, z0, weight, bias):
        lib = _lib.load()
        out = torch.empty(self.B, self.n_out, self.H, dtype=self.dtype, device=self.device)
        n_stage = 4 * max(self.grid.numel() - 1, 0)
        self.stage_index = torch.empty(max(n_stage, 1), dtype=torch.int64, device=self.device)
        self.stage_frac = torch.empty(max(n_stage, 1), dtype=self.dtype, device=self.device)
        z0c = z0.detach().reshape(self.B, self.H).contiguous()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous()
        begin = self._mark()
        if self.method != _lib.METHOD_RK4:
            _lib.check(lib.cde_fixed_forward_linear(
                self.method, _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w),
                _lib.ptr(b), _lib.ptr(z0c), _lib.ptr(self.grid), self.grid.numel(), _lib.ptr(self.t_out), self.n_out,
                _lib.ptr(out), self.B, self.C, self.H, _lib.dtype_enum(self.dtype), _lib.dtype_enum(self.time_dtype),
                _lib.ptr(self.stage_index), _lib.ptr(self.stage_frac), _lib.stream_ptr(self.device)),
                "cde_fixed_forward_linear")
        else:
            _lib.check(lib.cde_rk4_forward_linear(
                _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b),
                self.act, _lib.ptr(z0c), _lib.ptr(self.grid), self.grid.numel(), _lib.ptr(self.t_out), self.n_out,
                _lib.ptr(out), self.B, self.C, self.H, _lib.dtype_enum(self.dtype), _lib.dtype_enum(self.time_dtype),
                self.variant, _lib.ptr(self.stage_index), _lib.ptr(self.stage_frac), _lib.stream_ptr(self.device)),
                "cde_rk4_forward_linear")
        if begin is not None:
            self.owner.event_log.append(("forward", begin, self._mark()))
        return out

    # adjoint=False: K2 that also stores every stage state, and the reverse-mode sweep over them (K3d)
    def run_forward_stages(self, z0, weight, bias):
        lib = _lib.load()
        out = torch.empty(self.B, self.n_out, self.H, dtype=self.dtype, device=self.device)
        n_steps = max(self.grid.numel() - 1, 0)
        self.stage_index = torch.empty(max(4 * n_steps, 1), dtype=torch.int64, device=self.device)
        self.stage_frac = torch.empty(max(4 * n_steps, 1), dtype=self.dtype, device=self.device)
        stages = torch.empty(self.B, max(n_steps, 1), 4, 32, dtype=torch.float32, device=self.device)
        z0c = z0.detach().reshape(self.B, self.H).contiguous()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous()
        begin = self._mark()
        _lib.check(lib.cde_rk4_forward_linear_stages(
            _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b), self.act,
            _lib.ptr(z0c), _lib.ptr(self.grid), self.grid.numel(), _lib.ptr(self.t_out), self.n_out, _lib.ptr(out),
            _lib.ptr(stages), self.B, self.C, self.H, _lib.dtype_enum(self.dtype), _lib.dtype_enum(self.time_dtype),
            _lib.ptr(self.stage_index), _lib.ptr(self.stage_frac), _lib.stream_ptr(self.device)),
            "cde_rk4_forward_linear_stages")
        if begin is not None:
            self.owner.event_log.append(("forward", begin, self._mark()))
        return out, stages

    def run_backprop(self, stages, grad_out, weight, bias, want_control=False):
        lib = _lib.load()
        step_dt, node_ptr, node_out, node_weight, n_steps = self.grids.backprop_lists()
        nbytes = lib.cde_rk4_backprop_workspace_bytes(self.B)
        workspace = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        grad_z0 = torch.empty(self.B, self.H, dtype=self.dtype, device=self.device)
        n_w = self.H * self.C * self.H
        flat = torch.empty(n_w + self.H * self.C, dtype=self.dtype, device=self.device)     # one buffer: see run_adjoint
        grad_w, grad_b = flat[:n_w].view(self.H * self.C, self.H), flat[n_w:]
        go = grad_out.detach().reshape(self.B, self.n_out, self.H).contiguous()
        w = weight.detach().contiguous()
        bvec = bias.detach().contiguous()
        begin = self._mark()
        grad_x = None
        if want_control:
            # the control's tensors require a gradient (autograd reaches them through X.derivative at every stage)
            grad_x = torch.zeros_like(self.coeffs)          # (B, rows, width) like the packed coefficients
            _lib.check(lib.cde_rk4_backprop_linear_dcontrol(
                _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(bvec),
                self.act, _lib.ptr(stages), _lib.ptr(go), self.n_out, _lib.ptr(step_dt), n_steps, _lib.ptr(node_ptr),
                _lib.ptr(node_out), _lib.ptr(node_weight), _lib.ptr(grad_z0), _lib.ptr(grad_w), _lib.ptr(grad_b),
                _lib.ptr(grad_x), self.B, self.C, self.H, _lib.dtype_enum(self.dtype), _lib.ptr(self.stage_index),
                _lib.ptr(self.stage_frac), _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr(self.device)),
                "cde_rk4_backprop_linear_dcontrol")
        else:
            _lib.check(lib.cde_rk4_backprop_linear(
                _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(bvec),
                self.act, _lib.ptr(stages), _lib.ptr(go), self.n_out, _lib.ptr(step_dt), n_steps, _lib.ptr(node_ptr),
                _lib.ptr(node_out), _lib.ptr(node_weight), _lib.ptr(grad_z0), _lib.ptr(grad_w), _lib.ptr(grad_b), self.B,
                self.C, self.H, _lib.dtype_enum(self.dtype), _lib.ptr(self.stage_index), _lib.ptr(self.stage_frac),
                _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr(self.device)), "cde_rk4_backprop_linear")
        if begin is not None:
            self.owner.event_log.append(("backprop", begin, self._mark()))
        return grad_z0, grad_w, grad_b, grad_x

    # backward: K3
    def run_adjoint(self, z_saved, grad_out, weight, bias, want_control=False):
        lib = _lib.load()
        sgrid, seg_off, n_sgrid = self.grids.sgrid, self.grids.seg_off, self.grids.n_sgrid
        dt = _lib.dtype_enum(self.dtype)
        variant = _lib.VARIANT_MFMA if want_control else self.variant
        fixed = self.method != _lib.METHOD_RK4
        nbytes = (lib.cde_fixed_adjoint_workspace_bytes(self.B, n_sgrid) if fixed else
                  lib.cde_rk4_adjoint_workspace_bytes(self.B, self.C, self.H, n_sgrid, dt, variant))
        workspace = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        grad_z0 = torch.empty(self.B, self.H, dtype=self.dtype, device=self.device)
        # weight and bias gradients are two views of ONE flat buffer: a data-parallel caller can all-reduce that
        # buffer in a single collective without gather/scatter copies (torchcde_amd.distributed.allreduce_gradients)
        n_w = self.H * self.C * self.H
        flat = torch.empty(n_w + self.H * self.C, dtype=self.dtype, device=self.device)
        grad_w, grad_b = flat[:n_w].view(self.H * self.C, self.H), flat[n_w:]
        zs = z_saved.detach().contiguous()
        go = grad_out.detach().reshape(self.B, self.n_out, self.H).contiguous()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous()
        begin = self._mark()
        grad_x = None
        if fixed:
            _lib.check(lib.cde_fixed_adjoint_linear(
                self.method, _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w),
                _lib.ptr(b), _lib.ptr(zs), _lib.ptr(go), _lib.ptr(sgrid), n_sgrid, _lib.ptr(seg_off), self.n_out,
                _lib.ptr(grad_z0), _lib.ptr(grad_w), _lib.ptr(grad_b), self.B, self.C, self.H, dt,
                _lib.dtype_enum(self.time_dtype), _lib.ptr(workspace), workspace.numel(), _lib.stream_ptr(self.device)),
                "cde_fixed_adjoint_linear")
        elif want_control:
            grad_x = torch.zeros_like(self.coeffs)          # (B, rows, width) like the packed coefficients
            _lib.check(lib.cde_rk4_adjoint_linear_dcontrol(
                _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b),
                self.act, _lib.ptr(zs), _lib.ptr(go), _lib.ptr(sgrid), n_sgrid, _lib.ptr(seg_off), self.n_out,
                _lib.ptr(grad_z0), _lib.ptr(grad_w), _lib.ptr(grad_b), _lib.ptr(grad_x), self.B, self.C, self.H, dt,
                _lib.dtype_enum(self.time_dtype), _lib.ptr(workspace), workspace.numel(),
                _lib.stream_ptr(self.device)), "cde_rk4_adjoint_linear_dcontrol")
        else:
            _lib.check(lib.cde_rk4_adjoint_linear(
                _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b),
                self.act, _lib.ptr(zs), _lib.ptr(go), _lib.ptr(sgrid), n_sgrid, _lib.ptr(seg_off),
                ctypes.cast(self.grids.seg_off_c, ctypes.c_void_p), self.n_out,
                _lib.ptr(grad_z0), _lib.ptr(grad_w), _lib.ptr(grad_b), self.B, self.C, self.H, dt,
                _lib.dtype_enum(self.time_dtype), self.variant, _lib.ptr(workspace), workspace.numel(),
                _lib.stream_ptr(self.device)), "cde_rk4_adjoint_linear")
        if begin is not None:
            self.owner.event_log.append(("adjoint", begin, self._mark()))
        return grad_z0, grad_w, grad_b, grad_x


def _output_layer_gradients(acc2, H, C, width):
    """(dL/dW2, dL/db2) from the reduced images: (halves, 256, 132), rows = the padded (hidden unit, channel) layout of the G2
    rows -- 32 units x 8 channels, or 16 x 16 per half (two halves: hidden units 0..15 and 16..31) -- bias in column 128."""
    units, channels = (32, 8) if C <= 8 else (16, 16)
    w = acc2[:, :, :width].reshape(acc2.size(0) * units, channels, width)[:H, :C].reshape(H * C, width)
    b = acc2[:, :, 128].reshape(acc2.size(0) * units, channels)[:H, :C].reshape(H * C)
    return w, b


class _MlpPlan:
    """Fused RK4 solves for the two-layer field: forward (K2m, cde_rk4_forward_mlp) and continuous-adjoint backward
    (K3m sweep + two library GEMMs, cde_rk4_adjoint_mlp_*)."""

    scratch_budget = 3 << 30        # bytes of HBM the backward sweep may use for its per-stage factors

    def __init__(self, path, field, batch, H, C, t, step_size, adjoint_step_size=None):
        coeffs, knots, _ = path._native_inputs()
        self.coeffs, self.knots = coeffs, knots
        self.n_intervals, self.degree = path._n_intervals(), path._degree
        self.field, self.batch, self.B, self.H, self.C = field, batch, coeffs.size(0), H, C
        self.path = path
        self.device = coeffs.device
        self.grids = _grids_for(_to_host(t), step_size, step_size if adjoint_step_size is None else adjoint_step_size,
                                self.device)
        self.n_out = self.grids.n_out

    def _weights(self, saved=None):
        f = self.field
        tensors = (f.hidden.weight, f.hidden.bias, f.output.weight, f.output.bias) if saved is None else saved
        return tuple(p.detach().contiguous() for p in tensors)

    def run(self, z0):
        lib = _lib.load()
        g, f = self.grids, self.field
        out = torch.empty(self.B, g.n_out, self.H, dtype=torch.float32, device=self.device)
        n_stage = 4 * max(g.grid.numel() - 1, 0)
        stage_index = torch.empty(max(n_stage, 1), dtype=torch.int64, device=self.device)
        stage_frac = torch.empty(max(n_stage, 1), dtype=torch.float32, device=self.device)
        z0c = z0.detach().reshape(self.B, self.H).contiguous()
        w1, b1, w2, b2 = self._weights()
        _lib.check(lib.cde_rk4_forward_mlp(
            _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1), _lib.ptr(b1),
            w1.size(0), _lib.ptr(w2), _lib.ptr(b2), f.act, _lib.ptr(z0c), _lib.ptr(g.grid), g.grid.numel(),
            _lib.ptr(g.t_out), g.n_out, _lib.ptr(out), self.B, self.C, self.H, _lib.dtype_enum(torch.float32),
            _lib.dtype_enum(g.time_dtype), _lib.ptr(stage_index), _lib.ptr(stage_frac), _lib.stream_ptr(self.device)),
            "cde_rk4_forward_mlp")
        return out.reshape(*self.batch, g.n_out, self.H)

    def run_adjoint(self, z_saved, grad_out, want_control=False, weights=None):
        """torchdiffeq's odeint_adjoint backward for this field: per output interval (last to first) the augmented
        state is integrated in reversed time by K3m in chunks of steps; each chunk's per-stage factors (in HBM) are
        reduced into the parameter gradients by the split-K MFMA reduction (cde_mlp_grad_reduce; the bias gradients are
        the column sums of the same factors).
        ``weights``: the parameter tensors saved by the forward pass (an in-place update between forward and backward
        then trips autograd's version check instead of silently using the new values)."""
        lib = _lib.load()
        g, f = self.grids, self.field
        B, H, C = self.B, self.H, self.C
        dev, f32 = self.device, _lib.dtype_enum(torch.float32)
        w1, b1, w2, b2 = self._weights(weights)
        width = w1.size(0)
        nbytes = lib.cde_rk4_adjoint_mlp_workspace_bytes(g.n_sgrid)
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream = _lib.stream_ptr(dev)
        tdt = _lib.dtype_enum(g.time_dtype)
        z_saved = z_saved.detach().reshape(B, self.n_out, H)
        grad_out = grad_out.detach().reshape(B, self.n_out, H)
        # the sweep integrates (y, a) IN PLACE: private copies -- for a single series `[:, -1]` is already contiguous and
        # .contiguous() would hand the kernel the caller's own output / gradient tensors
        y = z_saved[:, -1].clone(memory_format=torch.contiguous_format)
        a = grad_out[:, -1].to(torch.float32).clone(memory_format=torch.contiguous_format)
        upper = _mlp_upper_half(H, C)     # 32 units x 16 channels: the sweep leaves the upper unit groups' dL/dY2 rows BEHIND
        halves = 2 if upper else 1        # the lower groups' in G2 (rows n .. 2n - 1); each half is reduced into its own image
        acc2 = torch.zeros(halves, 256, 132, dtype=torch.float32, device=dev)
        acc1 = torch.zeros(128, 36, dtype=torch.float32, device=dev)
        grad_x = torch.zeros_like(self.coeffs) if want_control else None     # accumulated by the sweep launches
        if g.n_sgrid > 1:
            _lib.check(lib.cde_rk4_adjoint_mlp_prepare(
                _lib.ptr(self.knots), self.n_intervals, _lib.ptr(g.sgrid), g.n_sgrid, _lib.ptr(w1), _lib.ptr(b1), width,
                _lib.ptr(w2), _lib.ptr(b2), C, H, f32, tdt, _lib.ptr(workspace), nbytes, stream),
                "cde_rk4_adjoint_mlp_prepare")
            row_bytes = (132 + 256 * halves + 128 + 36) * 4
            longest = max(g.seg_off_host[p + 1] - 1 - g.seg_off_host[p] for p in range(self.n_out - 1))
            chunk = max(1, min(longest, self.scratch_budget // (row_bytes * 4 * B)))
            rows = chunk * 4 * B
            U = torch.zeros(rows, 132, dtype=torch.float32, device=dev)
            U[:, 128] = 1
            Z = torch.zeros(rows, 36, dtype=torch.float32, device=dev)
            Z[:, 32] = 1
            G2 = torch.empty(halves * rows, 256, dtype=torch.float32, device=dev)
            G1 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
            reduce_ws = torch.empty(lib.cde_mlp_grad_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
        for p in range(self.n_out - 1):
            k, k_end = g.seg_off_host[p], g.seg_off_host[p + 1] - 1
            while k < k_end:
                ke = min(k_end, k + chunk)
                _lib.check(lib.cde_rk4_adjoint_mlp_sweep(
                    _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, f.act, _lib.ptr(y),
                    _lib.ptr(a), _lib.ptr(g.sgrid), g.n_sgrid, k, ke, _lib.ptr(U), _lib.ptr(G2), _lib.ptr(G1),
                    _lib.ptr(Z), _lib.ptr(grad_x), B, C, H, f32, tdt, _lib.ptr(workspace), nbytes, stream),
                    "cde_rk4_adjoint_mlp_sweep")
                n = 4 * (ke - k) * B
                for half in range(halves):
                    _lib.check(lib.cde_mlp_grad_reduce(_lib.ptr(G2[half * n:]), _lib.ptr(U), n, 2, _lib.ptr(acc2[half]),
                                                       _lib.ptr(reduce_ws), reduce_ws.numel(), stream), "cde_mlp_grad_reduce")
                _lib.check(lib.cde_mlp_grad_reduce(_lib.ptr(G1), _lib.ptr(Z), n, 1, _lib.ptr(acc1), _lib.ptr(reduce_ws),
                                                   reduce_ws.numel(), stream), "cde_mlp_grad_reduce")
                k = ke
            i_out = self.n_out - 1 - p
            y.copy_(z_saved[:, i_out - 1])                 # torchdiffeq: re-seed z from the stored forward solution
            a += grad_out[:, i_out - 1]                    # and add the incoming gradient at that output time
        grad_w2, grad_b2 = _output_layer_gradients(acc2, H, C, width)
        grad_w1 = acc1[:width, :H].contiguous()
        grad_b1 = acc1[:width, 32].contiguous()
        return a.reshape(*self.batch, H), grad_w1, grad_b1, grad_w2, grad_b2, grad_x

    # adjoint=False: K2m that also stores every stage state, and K3m's sweep as reverse mode through the steps
    def run_with_stages(self, z0):
        lib = _lib.load()
        g, f = self.grids, self.field
        out = torch.empty(self.B, g.n_out, self.H, dtype=torch.float32, device=self.device)
        n_steps = max(g.grid.numel() - 1, 0)
        stage_index = torch.empty(max(4 * n_steps, 1), dtype=torch.int64, device=self.device)
        stage_frac = torch.empty(max(4 * n_steps, 1), dtype=torch.float32, device=self.device)
        stages = torch.empty(self.B, max(n_steps, 1), 4, 32, dtype=torch.float32, device=self.device)
        z0c = z0.detach().reshape(self.B, self.H).contiguous()
        w1, b1, w2, b2 = self._weights()
        _lib.check(lib.cde_rk4_forward_mlp_stages(
            _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1), _lib.ptr(b1),
            w1.size(0), _lib.ptr(w2), _lib.ptr(b2), f.act, _lib.ptr(z0c), _lib.ptr(g.grid), g.grid.numel(),
            _lib.ptr(g.t_out), g.n_out, _lib.ptr(out), _lib.ptr(stages), self.B, self.C, self.H,
            _lib.dtype_enum(torch.float32), _lib.dtype_enum(g.time_dtype), _lib.ptr(stage_index), _lib.ptr(stage_frac),
            _lib.stream_ptr(self.device)), "cde_rk4_forward_mlp_stages")
        return out.reshape(*self.batch, g.n_out, self.H), stages

    def run_backprop(self, stages, grad_out, weights, want_control=False):
        """Reverse mode through the 3/8-rule steps (the gradient `loss.backward()` through torchdiffeq's own operations
        gives, reference solver.py:144 with adjoint=False): the sweep walks the forward grid backwards in chunks of steps
        that end on the grid nodes an output gradient lands on (the transpose of torchdiffeq's linear output
        interpolation, `_Grids.backprop_lists`); each chunk's factor rows go through the same split-K reduction as the
        continuous adjoint's."""
        lib = _lib.load()
        g, f = self.grids, self.field
        B, H, C = self.B, self.H, self.C
        dev, f32 = self.device, _lib.dtype_enum(torch.float32)
        w1, b1, w2, b2 = self._weights(weights)
        width = w1.size(0)
        _, _, _, _, n_steps = g.backprop_lists()
        nodes = g.backprop_nodes
        n_grid = n_steps + 1
        go = grad_out.detach().reshape(B, self.n_out, H).to(torch.float32)
        gy = torch.zeros(B, H, dtype=torch.float32, device=dev)

        def land(m):                                                # output gradients that land on grid node m
            for j, wgt in nodes[m]:
                gy.add_(go[:, j], alpha=wgt)
        land(n_steps)
        upper = _mlp_upper_half(H, C)     # 32 units x 16 channels: the sweep leaves the upper unit groups' dL/dY2 rows BEHIND
        halves = 2 if upper else 1        # the lower groups' in G2 (rows n .. 2n - 1); each half is reduced into its own image
        acc2 = torch.zeros(halves, 256, 132, dtype=torch.float32, device=dev)
        acc1 = torch.zeros(128, 36, dtype=torch.float32, device=dev)
        grad_x = torch.zeros_like(self.coeffs) if want_control else None     # accumulated by the sweep launches
        if n_steps > 0:
            stream = _lib.stream_ptr(dev)
            tdt = _lib.dtype_enum(g.time_dtype)
            nbytes = lib.cde_rk4_adjoint_mlp_workspace_bytes(n_grid)
            workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.cde_rk4_backprop_mlp_prepare(
                _lib.ptr(self.knots), self.n_intervals, _lib.ptr(g.grid), n_grid, _lib.ptr(w1), _lib.ptr(b1), width,
                _lib.ptr(w2), _lib.ptr(b2), C, H, f32, tdt, _lib.ptr(workspace), nbytes, stream),
                "cde_rk4_backprop_mlp_prepare")
            row_bytes = (132 + 256 * halves + 128 + 36) * 4
            chunk = max(1, min(n_steps, self.scratch_budget // (row_bytes * 4 * B)))
            rows = chunk * 4 * B
            U = torch.zeros(rows, 132, dtype=torch.float32, device=dev)
            U[:, 128] = 1
            Z = torch.zeros(rows, 36, dtype=torch.float32, device=dev)
            Z[:, 32] = 1
            G2 = torch.empty(halves * rows, 256, dtype=torch.float32, device=dev)
            G1 = torch.empty(rows, 128, dtype=torch.float32, device=dev)
            reduce_ws = torch.empty(lib.cde_mlp_grad_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
            k_hi = n_steps
            while k_hi > 0:
                k_lo = max(0, k_hi - chunk)
                for m in range(k_hi - 1, k_lo, -1):                 # stop on the highest node below k_hi that receives a gradient
                    if nodes[m]:
                        k_lo = m
                        break
                if want_control:
                    _lib.check(lib.cde_rk4_backprop_mlp_sweep_dcontrol(
                        _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, f.act, _lib.ptr(stages),
                        _lib.ptr(gy), _lib.ptr(g.grid), n_grid, k_lo, k_hi, _lib.ptr(U), _lib.ptr(G2), _lib.ptr(G1),
                        _lib.ptr(Z), _lib.ptr(grad_x), B, C, H, f32, tdt, _lib.ptr(workspace), nbytes, stream),
                        "cde_rk4_backprop_mlp_sweep_dcontrol")
                else:
                    _lib.check(lib.cde_rk4_backprop_mlp_sweep(
                        _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, f.act, _lib.ptr(stages),
                        _lib.ptr(gy), _lib.ptr(g.grid), n_grid, k_lo, k_hi, _lib.ptr(U), _lib.ptr(G2), _lib.ptr(G1),
                        _lib.ptr(Z), B, C, H, f32, tdt, _lib.ptr(workspace), nbytes, stream), "cde_rk4_backprop_mlp_sweep")
                n = 4 * (k_hi - k_lo) * B
                for half in range(halves):
                    _lib.check(lib.cde_mlp_grad_reduce(_lib.ptr(G2[half * n:]), _lib.ptr(U), n, 2, _lib.ptr(acc2[half]),
                                                       _lib.ptr(reduce_ws), reduce_ws.numel(), stream), "cde_mlp_grad_reduce")
                _lib.check(lib.cde_mlp_grad_reduce(_lib.ptr(G1), _lib.ptr(Z), n, 1, _lib.ptr(acc1), _lib.ptr(reduce_ws),
                                                   reduce_ws.numel(), stream), "cde_mlp_grad_reduce")
                land(k_lo)
                k_hi = k_lo
        else:
            pass                                                    # a single output time: node 0 is node n_steps, landed above
        grad_w2, grad_b2 = _output_layer_gradients(acc2, H, C, width)
        grad_w1 = acc1[:width, :H].contiguous()
        grad_b1 = acc1[:width, 32].contiguous()
        return gy.reshape(*self.batch, H), grad_w1, grad_b1, grad_w2, grad_b2, grad_x

    def time_gradients(self, z_saved, grad_out, weights, grad_x, t, want_t=True, want_knots=False):
        """Time gradients from what the sweep produced (as _plan_time_gradients for the one-layer fields):
        dL/dt_i = f(t_i, z_i) . dL/dz_i for i >= 1, dL/dt_0 = int a^T F(z) d2X/dt2 dt - sum_i dL/dt_i; the integral is a
        contraction of the control gradient with the cubic's (2c, 3d) rows and vanishes for a piecewise-linear control;
        dL/d knot_j = - its part over interval j (cubic), or through the widths of a piecewise-linear control."""
        B, H, C, f = self.B, self.H, self.C, self.field
        co = self.coeffs
        if self.degree == _lib.PATH_CUBIC:
            per_interval = (co[..., 2 * C:3 * C] * grad_x[..., C:2 * C]
                            + 2 * co[..., 3 * C:] * grad_x[..., 2 * C:3 * C]).sum(-1).sum(0)
        else:
            per_interval = torch.zeros(co.size(1) - 1, dtype=torch.float32, device=self.device)
        integral = per_interval.sum()
        grad_knots = None
        if want_knots and self.degree == _lib.PATH_CUBIC:
            grad_knots = torch.cat([-per_interval, per_interval.new_zeros(1)])
        elif want_knots:
            knots, values = self.knots.double(), co.double()
            slopes = (values[:, 1:] - values[:, :-1]) / (knots[1:] - knots[:-1]).unsqueeze(-1)
            dh = (grad_x.double().cumsum(1)[:, :-1] * slopes).sum((0, 2))            # dL/dh_j
            zero = dh.new_zeros(1)
            grad_knots = (torch.cat([zero, dh]) - torch.cat([dh, zero])).to(co.dtype)
        if not want_t:
            return None, grad_knots
        w1, b1, w2, b2 = self._weights(weights)
        zs = z_saved.detach().reshape(B, self.n_out, H)
        go = grad_out.detach().reshape(B, self.n_out, H)
        vals = [None] * self.n_out
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        for i in range(1, self.n_out):
            pre = torch.nn.functional.linear(torch.nn.functional.linear(zs[:, i], w1, b1).relu(), w2, b2)
            if f.act == _lib.ACT_TANH:
                pre = pre.tanh()
            dX = self.path.derivative(self.grids.t_out[i]).reshape(B, C)
            vals[i] = ((pre.view(B, H, C) * dX.unsqueeze(1)).sum(-1) * go[:, i]).sum()
            total = total + vals[i]
        vals[0] = integral - total
        grad_t = torch.stack(vals).to(device=t.device, dtype=t.dtype) if self.n_out > 1 else torch.zeros_like(t)
        return grad_t, grad_knots


class _FusedMlpRK4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z0, w1, b1, w2, b2, plan, want_x, t, knots, *control):
        out = plan.run(z0)
        ctx.plan, ctx.want_x, ctx.t_like, ctx.has_knots = plan, want_x, t, knots is not None
        ctx.save_for_backward(out, w1, b1, w2, b2)
        return out

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        out, *weights = ctx.saved_tensors
        plan = ctx.plan
        need = ctx.needs_input_grad
        want_t = ctx.t_like is not None and need[7]
        want_knots = ctx.has_knots and need[8]
        grad_z0, gw1, gb1, gw2, gb2, grad_x = plan.run_adjoint(out, grad_out, ctx.want_x, weights)
        grad_t = grad_knots = None
        if want_t or want_knots:
            if grad_x is None and (plan.degree == _lib.PATH_CUBIC or want_knots):
                # the time terms come out of the control-gradient sweep: run IN ADDITION, so that the other gradients are
                # bitwise the same whether or not a time requires grad (the reference's "detach trick" invariance,
                # test/test_tricks.py:111-131), as _FusedRK4 does
                grad_x = plan.run_adjoint(out, grad_out, True, weights)[5]
            grad_t, grad_knots = plan.time_gradients(out, grad_out, weights, grad_x, ctx.t_like, want_t, want_knots)
            grad_knots = _with_fit_chain(plan, grad_knots, grad_x)
        control_grads = ()
        if ctx.want_x:
            C = plan.C
            gx = grad_x.reshape(*plan.batch, grad_x.size(-2), grad_x.size(-1))
            pieces = (gx[..., C:2 * C], gx[..., 2 * C:3 * C], gx[..., 3 * C:]) if plan.degree == _lib.PATH_CUBIC else (gx,)
            control_grads = tuple(g if n else None for g, n in zip(pieces, need[9:]))
        return (grad_z0 if need[0] else None, gw1 if need[1] else None, gb1 if need[2] else None,
                gw2 if need[3] else None, gb2 if need[4] else None, None, None, grad_t, grad_knots) + control_grads


def _mlp_upper_half(H, C):
    """32 hidden units x 16 channels (config 5 at hidden size 32: 14 logsignature channels): twice the 16 tiles -- the kernels
    read the upper unit groups straight from the output layer's tensors (csrc/cde_mfma.h: MlpHi)."""
    return 16 < H <= 32 and 8 < C <= 16


# which requests of the 32 x 16 shape the kernels take (the rest of that shape is solved step-wise)
# (the adaptive backward: one GPU's batch -- its shared-controller protocol exchanges one layer-2 image, not two)
_UPPER_HALF_PATHS = {"mlp_rk4_forward", "mlp_dopri5_forward", "mlp_rk4_adjoint", "mlp_rk4_backprop", "mlp_dopri5_adjoint"}


def _mlp_fusable(field, H, C, z0, packed):
    w1, w2 = field.hidden.weight, field.output.weight
    # tiles of the two-layer kernels: 32 hidden units x 8 channels, or 16 x 16 (cde_mi355x.h: cde_rk4_forward_mlp); 32 x 16 with
    # the upper half read from the raw tensors (16-byte rows: width a multiple of 4, an aligned contiguous weight)
    upper = (_mlp_upper_half(H, C) and w1.size(0) % 4 == 0 and w2.is_contiguous() and w2.data_ptr() % 16 == 0
             and field.output.bias.is_contiguous())
    return (z0.dtype == packed.dtype == w1.dtype == w2.dtype == torch.float32
            and ((H <= 32 and C <= 8) or (H <= 16 and C <= 16) or upper)
            and w1.size(0) <= 128 and tuple(w1.shape) == (w1.size(0), H) and tuple(w2.shape) == (H * C, w1.size(0)))


class _FusedRK4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z0, weight, bias, plan, wants, t, knots, *control):
        # `control`: the path's differentiable buffer views (cubic: b, 2c, 3d; linear: the knot values) when the
        # gradient w.r.t. the coefficients was requested through adjoint_params -- only their gradient slots are used.
        # `t` / `knots`: the output times / the path's knot times when THEIR gradient is wanted (else None).
        out = plan.run_forward(z0, weight, bias)
        ctx.plan, ctx.wants = plan, wants
        ctx.has_t, ctx.has_knots = t is not None, knots is not None
        ctx.save_for_backward(out, weight, bias, *((t,) if t is not None else ()))
        return out.reshape(*plan.batch, plan.n_out, plan.H)

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        plan = ctx.plan
        if not plan.adjoint:
            raise NotImplementedError(
                "torchcde_amd: backpropagating through the solver's internal operations (adjoint=False) is not "
                "implemented on the native path; use adjoint=True (continuous adjoint, the reference's default).")
        z_saved, weight, bias, *rest = ctx.saved_tensors
        want_w, want_b, want_x = ctx.wants
        want_t = ctx.has_t and ctx.needs_input_grad[5]
        want_knots = ctx.has_knots and ctx.needs_input_grad[6]
        times = want_t or want_knots
        grad_z0, grad_w, grad_b, grad_x = plan.run_adjoint(z_saved, grad_out, weight, bias, want_x)
        grad_t = grad_knots = None
        if times:
            if grad_x is None:
                # The time terms come out of the control-gradient sweep.  It is run IN ADDITION to the sweep above, so
                # that dL/dz0, dL/dW, dL/db are bitwise the same whether or not a time requires grad (the reference's
                # "detach trick" invariance, test/test_tricks.py:111-131); requesting time gradients is rare.
                grad_x = plan.run_adjoint(z_saved, grad_out, weight, bias, True)[3]
            grad_t, grad_knots = plan.time_gradients(z_saved, grad_out, weight, bias, grad_x, rest[0] if ctx.has_t else None,
                                                     want_t, want_knots)
            grad_knots = _with_fit_chain(plan, grad_knots, grad_x)
        control_grads = ()
        if want_x:
            C = plan.C
            gx = grad_x.reshape(*plan.batch, grad_x.size(-2), grad_x.size(-1))
            pieces = (gx[..., C:2 * C], gx[..., 2 * C:3 * C], gx[..., 3 * C:]) if plan.degree == _lib.PATH_CUBIC else (gx,)
            control_grads = tuple(g if need else None for g, need in zip(pieces, ctx.needs_input_grad[7:]))
        return (grad_z0.reshape(*plan.batch, plan.H) if ctx.needs_input_grad[0] else None,
                grad_w.view_as(weight) if (ctx.needs_input_grad[1] and want_w) else None,
                grad_b.view_as(bias) if (ctx.needs_input_grad[2] and want_b) else None,
                None, None, grad_t, grad_knots) + control_grads


def _reaches(tensor, leaf):
    """Does the autograd graph of `tensor` reach the leaf `leaf`?  (The coefficient tensor of a control fitted with the same
    knot tensor the spline is evaluated on: test/test_tricks.py:21-49.)"""
    start = getattr(tensor, "grad_fn", None)
    if start is None:
        return False
    target = leaf.grad_fn                       # (a knot tensor that is itself computed: its own node is what the graph reaches)
    seen, stack = set(), [start]
    while stack:
        node = stack.pop()
        if node is None or node in seen:
            continue
        seen.add(node)
        if (target is not None and node is target) or (target is None and getattr(node, "variable", None) is leaf):
            return True
        stack.extend(fn for fn, _ in node.next_functions)
    return False


def _knot_fit_chain(X):
    """The path's buffers whose autograd graph reaches the knot tensor `X._t` (and that tensor), or None.
    torchdiffeq's adjoint differentiates the vector field w.r.t. every entry of adjoint_params with `torch.autograd.grad`
    (its `vjp_params`); for the knot tensor that derivative runs through EVERYTHING the field evaluation reads that was
    computed from it -- the spline's `frac = t - t_j`, and, when the coefficients were fitted with the same tensor
    (`natural_cubic_coeffs(x, t)` then `CubicSpline(coeffs, t)`, as the reference's test does), the fit as well.  The
    fused kernels form the first term; the second is one vector-Jacobian product of the fit with dL/dcoeffs (the block
    integrals are linear in it), added on the host (_with_fit_chain).  With the coefficient tensor in adjoint_params too
    the reference thus counts the fit's chain twice -- once inside the knot block, once when the returned dL/dcoeffs flows
    back through the fit -- and so does this path (the float64 oracle, built on autograd the same way, shows the same)."""
    if not (isinstance(X._t, torch.Tensor) and X._t.requires_grad):
        return None
    buffers = tuple(b for b in X._control_buffers() if _reaches(b, X._t))
    return (buffers, tuple(X._control_buffers()), X._t) if buffers else None


def _with_fit_chain(plan, grad_knots, grad_x):
    """dL/d(knot times) plus the fit's chain (see _knot_fit_chain); `grad_x`: dL/d(packed coefficients) of the sweep."""
    chain = getattr(plan, "fit_chain", None)
    if chain is None or grad_knots is None or grad_x is None:
        return grad_knots
    reaching, buffers, knots = chain
    C = plan.C
    gx = grad_x.reshape(*plan.batch, grad_x.size(-2), grad_x.size(-1))
    pieces = (gx[..., C:2 * C], gx[..., 2 * C:3 * C], gx[..., 3 * C:]) if plan.degree == _lib.PATH_CUBIC else (gx,)
    pairs = [(b, g.to(b.dtype)) for b, g in zip(buffers, pieces) if any(b is r for r in reaching)]
    (extra,) = torch.autograd.grad([b for b, _ in pairs], [knots], [g.contiguous() for _, g in pairs], retain_graph=True,
                                   allow_unused=True)
    return grad_knots if extra is None else grad_knots + extra.to(device=grad_knots.device, dtype=grad_knots.dtype)


def _control_gradients(plan, grad_x, want_x, want_knots, need, knot_gradient):
    """(dL/d knot times, *dL/d the path's buffer views) of a reverse-mode sweep: the sweep leaves dL/d(packed coefficients)
    in `grad_x`; the knot times follow from it by the chain rule of `frac = t - t_j` / of the widths (plan.time_gradients)."""
    grad_knots = knot_gradient() if want_knots else None
    if not want_x:
        return (grad_knots,)
    C = plan.C
    gx = grad_x.reshape(*plan.batch, grad_x.size(-2), grad_x.size(-1))
    pieces = (gx[..., C:2 * C], gx[..., 2 * C:3 * C], gx[..., 3 * C:]) if plan.degree == _lib.PATH_CUBIC else (gx,)
    return (grad_knots,) + tuple(g if n else None for g, n in zip(pieces, need))


class _FusedMlpRK4Backprop(torch.autograd.Function):
    """cdeint(..., method='rk4', adjoint=False) for the examples' two-layer field: as _FusedRK4Backprop, on K2m / K3m."""

    @staticmethod
    def forward(ctx, z0, w1, b1, w2, b2, plan, want_x, knots, *control):
        # `control` / `knots`: the path's differentiable buffer views / its knot times when autograd must reach them (with
        # adjoint=False it does so through X.derivative at every stage) -- only their gradient slots are used
        out, stages = plan.run_with_stages(z0)
        ctx.plan, ctx.want_x, ctx.has_knots = plan, want_x, knots is not None
        ctx.save_for_backward(stages, w1, b1, w2, b2)
        return out

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        stages, *weights = ctx.saved_tensors
        plan, need = ctx.plan, ctx.needs_input_grad
        want_knots = ctx.has_knots and need[7]
        grad_z0, g1w, g1b, g2w, g2b, grad_x = plan.run_backprop(stages, grad_out, weights, ctx.want_x or want_knots)
        return (grad_z0 if need[0] else None, g1w if need[1] else None, g1b if need[2] else None,
                g2w if need[3] else None, g2b if need[4] else None, None, None) + _control_gradients(
                    plan, grad_x, ctx.want_x, want_knots, need[8:], lambda: plan.time_gradients(
                        None, None, weights, grad_x, None, False, True)[1])


class _FusedRK4Backprop(torch.autograd.Function):
    """cdeint(..., method='rk4', adjoint=False) for the affine field: torchdiffeq.odeint under autograd (reference
    solver.py:226-227 with adjoint=False) -- the forward kernel stores the state handed to every field evaluation, the
    backward is reverse-mode differentiation of the 3/8-rule steps themselves (csrc/rk4_backprop.hip), i.e. the gradient
    of the discrete solve, as `loss.backward()` through torchdiffeq's own operations gives it."""

    @staticmethod
    def forward(ctx, z0, weight, bias, plan, want_x, knots, *control):
        # `control` / `knots`: as for _FusedMlpRK4Backprop
        out, stages = plan.run_forward_stages(z0, weight, bias)
        ctx.plan, ctx.want_x, ctx.has_knots = plan, want_x, knots is not None
        ctx.save_for_backward(weight, bias, stages)
        return out.reshape(*plan.batch, plan.n_out, plan.H)

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        plan, need = ctx.plan, ctx.needs_input_grad
        weight, bias, stages = ctx.saved_tensors
        want_knots = ctx.has_knots and need[5]
        grad_z0, grad_w, grad_b, grad_x = plan.run_backprop(stages, grad_out, weight, bias, ctx.want_x or want_knots)
        return (grad_z0.reshape(*plan.batch, plan.H) if need[0] else None,
                grad_w.view_as(weight) if need[1] else None,
                grad_b if need[2] else None, None, None) + _control_gradients(
                    plan, grad_x, ctx.want_x, want_knots, need[6:], lambda: plan.time_gradients(
                        None, None, weight, bias, grad_x, None, False, True)[1])


# ------------------------------------------------------------------------------------------ dopri5 (K4)
last_dispatch = dispatch.last      # () -> (Choice(path, reason), Request) of this thread's most recent cdeint call
# Module attributes served from the calling thread's _CallState (see the module class at the end of this file):
#   last_dopri5_stats          {"n_accept", "n_reject", "launches"} of the most recent adaptive solve (tests / logging)
#   last_dopri5_adjoint_stats  the same for the most recent fused adaptive backward (summed over the output intervals)
#   record_dopri5_steps        tests: also fetch the accepted (t0, t1, on_jump) steps into last_dopri5_stats["steps"]
#   event_log                  bench.py: a list that receives the HIP events around the rk4 C-ABI calls
_DOPRI_CHUNK = 48          # attempt kernels queued between two looks at the done flag
_WORKSPACE_HEAD = 1 << 20  # bytes at the start of an adaptive workspace that hold the controller blocks and partial sums


class _Dopri5Plan:
    def __init__(self, path, field, batch, H, C, t, rtol, atol, options, variant=_lib.VARIANT_AUTO,
                 adjoint_rtol=None, adjoint_atol=None, adjoint_options=None):
        self.owner = _state()
        self.record = self.owner.record
        self.variant = variant
        self.adjoint_rtol = float(rtol if adjoint_rtol is None else adjoint_rtol)
        self.adjoint_atol = float(atol if adjoint_atol is None else adjoint_atol)
        from .distributed import step_control
        self.shared = step_control()      # one controller across the shards of a distributed batch (or None)
        options = {} if options is None else dict(options)
        jump_t = options.pop("jump_t", None)
        self.safety = float(options.pop("safety", 0.9))
        self.ifactor = float(options.pop("ifactor", 10.0))
        self.dfactor = float(options.pop("dfactor", 0.2))
        for key in ("first_step", "step_t", "min_step", "max_step", "max_num_steps", "dtype", "norm"):
            if options.get(key, None) is not None:
                raise NotImplementedError("torchcde_amd: dopri5 option %r is not supported natively" % key)
            options.pop(key, None)
        if options:
            raise NotImplementedError("torchcde_amd: unsupported dopri5 options %s" % sorted(options))
        coeffs, knots, _ = path._native_inputs()
        self.coeffs, self.knots = coeffs, knots
        self.path = path
        self.n_intervals, self.degree, self.act = path._n_intervals(), path._degree, field.act
        self.hidden = field.hidden if field.kind == "mlp2" else None      # two-layer field: its first Linear
        self.batch, self.B, self.H, self.C = batch, coeffs.size(0), H, C
        self.dtype, self.device = coeffs.dtype, coeffs.device
        self.rtol = float(rtol)
        self.atol = float(atol)
        t_host = _to_host(t).to(torch.float64)
        self.n_out = t_host.numel()
        self.t_host = t_host
        self.t_out = t_host.to(self.device)
        if jump_t is None:
            self.jump_t, self.n_jump, self.jump_s = None, 0, None
        else:
            jt = _to_host(torch.as_tensor(jump_t)).to(torch.float64).reshape(-1)
            jt = torch.sort(jt).values
            self.jump_t, self.n_jump = jt.to(self.device), jt.numel()
            self.jump_s = (-jt).flip(0).contiguous().to(self.device)      # the backward solve runs in s = -t
        self.n_jump_s = self.n_jump
        self.adj_safety, self.adj_ifactor, self.adj_dfactor = self.safety, self.ifactor, self.dfactor
        self.adj_norm_kind = 0      # torchdiffeq's default adjoint norm: mixed over (vjp_t, y, a, parameter gradients)
        if adjoint_options is not None:
            # torchdiffeq: explicit adjoint_options REPLACE the forward options for the backward solve (no jump_t unless
            # it is repeated there); norm: absent (the default mixed norm) or "seminorm" (without the parameter blocks)
            adj = dict(adjoint_options)
            self.adj_norm_kind = 1 if adj.pop("norm", None) == "seminorm" else 0
            jump_b = adj.pop("jump_t", None)
            self.adj_safety = float(adj.pop("safety", 0.9))
            self.adj_ifactor = float(adj.pop("ifactor", 10.0))
            self.adj_dfactor = float(adj.pop("dfactor", 0.2))
            if jump_b is None:
                self.jump_s, self.n_jump_s = None, 0
            else:
                jb = torch.sort(_to_host(torch.as_tensor(jump_b)).to(torch.float64).reshape(-1)).values
                self.jump_s, self.n_jump_s = (-jb).flip(0).contiguous().to(self.device), jb.numel()

    def _run_shared(self, lib, shared, out, z0c, w, b, dt, workspace, w1=None, b1=None):
        """One controller for all shards: per attempted step the pending error sums are all-reduced before the launch
        that consumes them (torchcde_amd.distributed.shared_step_control)."""
        reduce, global_batch = shared
        sums = torch.zeros(2, dtype=torch.float64, device=self.device)
        size = ctypes.sizeof(_lib.DopriStatus)
        stream = _lib.stream_ptr(self.device)
        launched = 0
        while True:
            for _ in range(_DOPRI_CHUNK):
                if self.hidden is None:
                    _lib.check(lib.cde_dopri5_pending_sums(_lib.ptr(workspace), workspace.numel(), self.B, self.C, self.H, dt,
                                                           self.variant, self.act, launched, _lib.ptr(sums), stream),
                               "cde_dopri5_pending_sums")
                    reduce(sums)
                    _lib.check(lib.cde_dopri5_advance_sharded(
                        _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b),
                        self.act, _lib.ptr(z0c), _lib.ptr(self.t_out), self.n_out, _lib.ptr(self.jump_t), self.n_jump,
                        self.rtol, self.atol, self.safety, self.ifactor, self.dfactor, _lib.ptr(out), self.B, self.C, self.H,
                        dt, self.variant, _lib.ptr(workspace), workspace.numel(), launched, _lib.ptr(sums), global_batch,
                        stream), "cde_dopri5_advance_sharded")
                else:                                                  # the two-layer field (round 4)
                    _lib.check(lib.cde_dopri5_pending_sums_mlp(_lib.ptr(workspace), workspace.numel(), self.B, self.C, self.H,
                                                               dt, launched, _lib.ptr(sums), stream),
                               "cde_dopri5_pending_sums_mlp")
                    reduce(sums)
                    _lib.check(lib.cde_dopri5_advance_mlp_sharded(
                        _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1), _lib.ptr(b1),
                        w1.size(0), _lib.ptr(w), _lib.ptr(b), self.act, _lib.ptr(z0c), _lib.ptr(self.t_out), self.n_out,
                        _lib.ptr(self.jump_t), self.n_jump, self.rtol, self.atol, self.safety, self.ifactor, self.dfactor,
                        _lib.ptr(out), self.B, self.C, self.H, dt, _lib.ptr(workspace), workspace.numel(), launched,
                        _lib.ptr(sums), global_batch, stream), "cde_dopri5_advance_mlp_sharded")
                launched += 1
            raw = workspace[(launched & 1) * size:(launched & 1) * size + size].cpu().numpy().tobytes()
            status = _lib.DopriStatus.from_buffer_copy(raw)
            if status.phase == 4:
                break
            if launched > 2_000_000:
                raise RuntimeError("torchcde_amd: dopri5 did not reach t[-1] after %d attempted steps" % launched)
        last_dopri5_stats = self.owner.dopri5            # (updated in place: callers may hold the dict itself)
        last_dopri5_stats.clear()
        last_dopri5_stats.update(n_accept=status.n_accept, n_reject=status.n_reject, launches=launched)
        if self.record:
            off = lib.cde_dopri5_trace_offset(self.B, self.C, self.H, dt)
            n = min(status.n_accept, 4096)
            last_dopri5_stats["steps"] = workspace[off:off + 24 * n].view(torch.float64).view(n, 3).cpu()
        return out

    def run_adjoint(self, z_saved, grad_out, weight, bias, want_t=False, want_control=False, want_knots=False):
        """K4a: torchdiffeq's odeint_adjoint backward for the adaptive solve -- default mixed norm (or "seminorm"), dense
        output at the interval ends -- one attempt kernel + one reduction kernel per attempted step
        (csrc/dopri5_adjoint.hip), output intervals from the last to the first.
        want_t: the output times require a gradient (torchdiffeq's time_vjps): dL/dt_i = f(t_i, z_i) . dL/dz_i for i >= 1
        (one field evaluation per output time, here); vjp_t -- which K4a integrates and measures in its error norm anyway --
        starts every interval at its carried value minus that term, and is dL/dt_0 after the last one.
        want_control: adjoint_params names the coefficient tensor (self.control_numel elements): dL/dcoeffs, packed layout,
        and (want_knots: the knot times are there too; else None) dL/d knots are appended to the result -- more blocks of
        the mixed norm, accumulated on the device (cde_dopri5_adjoint_advance_dcontrol)."""
        lib = _lib.load()
        B, H, C, dev = self.B, self.H, self.C, self.device
        z_saved = z_saved.detach().reshape(B, self.n_out, H)
        grad_out = grad_out.detach().reshape(B, self.n_out, H).to(torch.float32)
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        n_w = H * C * H
        flat = torch.zeros(n_w + H * C, dtype=torch.float32, device=dev)       # one buffer: a single all-reduce upstream
        grad_w, grad_b = flat[:n_w].view(H * C, H), flat[n_w:]
        a = grad_out[:, -1].contiguous()
        grad_x = torch.zeros_like(self.coeffs) if want_control else None
        grad_k = torch.zeros_like(self.knots) if (want_control and want_knots) else None
        tail = (grad_x, grad_k) if want_control else ()
        if self.n_out == 1:
            return ((a, grad_w, grad_b, torch.zeros(1, dtype=torch.float32, device=dev)) if want_t else (a, grad_w, grad_b)) + tail
        if want_control and self.shared is not None:
            raise NotImplementedError("torchcde_amd: control gradients through the adaptive backward have no shared-controller form")
        nbytes = (lib.cde_dopri5_adjoint_dcontrol_workspace_bytes if want_control else lib.cde_dopri5_adjoint_workspace_bytes)(B, C, H)
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        workspace[:_WORKSPACE_HEAD].zero_()     # controller blocks + partial sums: defined before the first launch reads them
        size = ctypes.sizeof(_lib.DopriStatus)
        stride = lib.cde_dopri5_adjoint_status_stride()
        a_out = torch.empty(B, H, dtype=torch.float32, device=dev)
        shared = self.shared
        reduced = None
        # "seminorm": the parameter blocks are not part of the decision -- only the 8 state sums travel between the shards
        state_only = shared is not None and self.adj_norm_kind == 1
        if shared is not None:
            reduced = torch.zeros(8 if state_only else lib.cde_dopri5_adjoint_reduced_count(), dtype=torch.float64, device=dev)
        stream = _lib.stream_ptr(dev)
        stats = dict(n_accept=0, n_reject=0, launches=0)
        steps, attempts = [], []
        time_terms = [None] * self.n_out
        carry = None
        if want_t:
            off = lib.cde_dopri5_adjoint_carry_offset(B, C, H)
            workspace[off:off + 256].zero_()                          # the whole carry block (the kernel leaves it alone: bit 1)
            carry = workspace[off:off + 8].view(torch.float64)        # vjp_t, carried across the intervals on the device
        if want_t:
            # torchdiffeq: func_eval = func(t[i], y[i]); dLd_cur_t = func_eval . grad_y[i]; aug_state[0] -= dLd_cur_t -- the
            # field at ALL output times in one batched evaluation before the loop
            pre = torch.nn.functional.linear(z_saved[:, 1:], w, b)                            # (B, T - 1, H * C)
            if self.act == _lib.ACT_TANH:
                pre = pre.tanh()
            dX = self.path.derivative(self.t_out[1:]).reshape(B, self.n_out - 1, C)
            f_all = (pre.view(B, self.n_out - 1, H, C) * dX.unsqueeze(2)).sum(-1)
            terms = (f_all * grad_out[:, 1:]).sum((0, 2))
            if shared is not None:
                # one controller for all shards: vjp_t is a quantity of the WHOLE batch (it is in the error norm), so the
                # terms it starts every interval from are summed over the shards; every rank returns the global dL/dt
                terms64 = terms.to(torch.float64)
                shared[0](terms64)
                terms = terms64.to(torch.float32)
            for i in range(1, self.n_out):
                time_terms[i] = terms[i - 1]
        for i in range(self.n_out - 1, 0, -1):
            y = z_saved[:, i].contiguous()
            s0, s1 = -float(self.t_host[i]), -float(self.t_host[i - 1])
            if want_t:
                carry.copy_((carry.to(torch.float32) - time_terms[i]).to(torch.float64))
            launched = 0
            while True:
                def advance(first, count, sums_ptr, global_batch):
                    head = (_lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w),
                            _lib.ptr(b), self.act, _lib.ptr(y), _lib.ptr(a), s0, s1, _lib.ptr(self.jump_s), self.n_jump_s,
                            self.adjoint_rtol, self.adjoint_atol, self.adj_safety, self.adj_ifactor, self.adj_dfactor,
                            self.adj_norm_kind, _lib.ptr(a_out), B, C, H, _lib.dtype_enum(torch.float32),
                            int(i == self.n_out - 1) | (2 if want_t else 0), _lib.ptr(workspace), workspace.numel(), first,
                            count)
                    if want_control:
                        _lib.check(lib.cde_dopri5_adjoint_advance_dcontrol(*head, _lib.ptr(grad_x), int(self.control_numel),
                                                                           _lib.ptr(grad_k), stream),
                                   "cde_dopri5_adjoint_advance_dcontrol")
                    else:
                        _lib.check(lib.cde_dopri5_adjoint_advance(*head, sums_ptr, global_batch, stream),
                                   "cde_dopri5_adjoint_advance")
                if shared is None:
                    advance(launched, _DOPRI_CHUNK, None, 0)
                    launched += _DOPRI_CHUNK
                else:
                    # one controller for all shards: per attempted step the state sums AND the gradient images of the
                    # attempt are all-reduced, then every shard commits / measures the parameter blocks on the same numbers
                    for _ in range(_DOPRI_CHUNK):
                        advance(launched, 1, _lib.ptr(reduced) if launched else None, shared[1])
                        launched += 1
                        if state_only:
                            _lib.check(lib.cde_dopri5_adjoint_state_sums(_lib.ptr(workspace), workspace.numel(), B, C, H,
                                                                         launched, _lib.ptr(reduced), stream),
                                       "cde_dopri5_adjoint_state_sums")
                            shared[0](reduced)
                            _lib.check(lib.cde_dopri5_adjoint_apply_state_sums(_lib.ptr(workspace), workspace.numel(), B, C,
                                                                               H, launched, _lib.ptr(reduced), stream),
                                       "cde_dopri5_adjoint_apply_state_sums")
                            continue
                        _lib.check(lib.cde_dopri5_adjoint_pending_sums(_lib.ptr(workspace), workspace.numel(), B, C, H,
                                                                       launched, _lib.ptr(reduced), stream),
                                   "cde_dopri5_adjoint_pending_sums")
                        shared[0](reduced)
                        _lib.check(lib.cde_dopri5_adjoint_apply_reduced(
                            _lib.ptr(workspace), workspace.numel(), B, C, H, self.adjoint_rtol, self.adjoint_atol, launched,
                            _lib.ptr(reduced), stream), "cde_dopri5_adjoint_apply_reduced")
                at = (launched & 1) * stride
                raw = workspace[at:at + size].cpu().numpy().tobytes()
                status = _lib.DopriStatus.from_buffer_copy(raw)
                if status.phase == 4:
                    break
                if launched > 2_000_000:
                    raise RuntimeError("torchcde_amd: the dopri5 adjoint did not reach t = %g after %d attempted steps"
                                       % (-s1, launched))
            stats["n_accept"] += status.n_accept
            stats["n_reject"] += status.n_reject
            stats["launches"] += launched
            if self.record:
                off = lib.cde_dopri5_adjoint_trace_offset(B, C, H)
                n = min(status.n_accept, 4096)
                steps.append(workspace[off:off + 24 * n].view(torch.float64).view(n, 3).cpu())
                off = lib.cde_dopri5_adjoint_attempt_trace_offset(B, C, H)
                n = min(status.n_accept + status.n_reject, 16384)
                attempts.append(workspace[off:off + 40 * n].view(torch.float64).view(n, 5).cpu())
            a = a_out + grad_out[:, i - 1]
        _lib.check(lib.cde_dopri5_adjoint_finish(_lib.ptr(workspace), workspace.numel(), _lib.ptr(grad_w), _lib.ptr(grad_b),
                                                 B, C, H, int(shared is not None and not state_only), stream),
                   "cde_dopri5_adjoint_finish")
        last_dopri5_adjoint_stats = self.owner.dopri5_adjoint
        last_dopri5_adjoint_stats.clear()
        last_dopri5_adjoint_stats.update(stats)
        if self.record:
            last_dopri5_adjoint_stats["steps"] = steps           # one (n, 3) tensor per output interval, last first
            last_dopri5_adjoint_stats["attempts"] = attempts     # (t0, t1, on_jump, accepted, ratio) of EVERY attempt
        if want_t:
            time_terms[0] = carry.to(torch.float32).reshape(())       # time_vjps[0] = the carried vjp_t
            return (a, grad_w, grad_b, torch.stack(time_terms)) + tail
        return (a, grad_w, grad_b) + tail

    def run_adjoint_mlp(self, z_saved, grad_out, w1, b1, w2, b2, want_t=False, want_control=False, want_knots=False):
        """K4am: the same backward for the two-layer field (csrc/dopri5_mlp_adjoint.hip): per attempted step the attempt
        kernel, the split-K reduction of its gradient factors and the commit / norm kernel, queued by one C-ABI call.
        want_t: output-time gradients, as in run_adjoint."""
        lib = _lib.load()
        B, H, C, dev = self.B, self.H, self.C, self.device
        z_saved = z_saved.detach().reshape(B, self.n_out, H)
        grad_out = grad_out.detach().reshape(B, self.n_out, H).to(torch.float32)
        w1, b1, w2, b2 = (p.detach().contiguous() for p in (w1, b1, w2, b2))
        width = w1.size(0)
        a = grad_out[:, -1].contiguous()
        # control gradients (as run_adjoint): dL/dcoeffs (packed layout) and, when the knot times are in adjoint_params too,
        # dL/d knots (else None) are appended to the result
        grad_x = torch.zeros_like(self.coeffs) if want_control else None
        grad_k = torch.zeros_like(self.knots) if (want_control and want_knots) else None
        tail = (grad_x, grad_k) if want_control else ()
        if self.n_out == 1:
            zeros = (a, torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2))
            return (zeros + (torch.zeros(1, dtype=torch.float32, device=dev),) if want_t else zeros) + tail
        if want_control and self.shared is not None:
            raise NotImplementedError("torchcde_amd: control gradients through the adaptive backward have no shared-controller form")
        nbytes = (lib.cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes if want_control else lib.cde_dopri5_adjoint_mlp_workspace_bytes)(B, C, H)
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        workspace[:_WORKSPACE_HEAD].zero_()
        size = ctypes.sizeof(_lib.DopriStatus)
        stride = lib.cde_dopri5_adjoint_status_stride()
        a_out = torch.empty(B, H, dtype=torch.float32, device=dev)
        stream = _lib.stream_ptr(dev)
        shared = self.shared
        reduced = None
        state_only = shared is not None and self.adj_norm_kind == 1      # "seminorm": only the 8 state sums travel
        if shared is not None:
            reduced = torch.zeros(8 if state_only else lib.cde_dopri5_adjoint_mlp_reduced_count(), dtype=torch.float64,
                                  device=dev)
        stats = dict(n_accept=0, n_reject=0, launches=0)
        steps, attempts = [], []
        time_terms = [None] * self.n_out
        carry = None
        if want_t:
            off = lib.cde_dopri5_adjoint_mlp_carry_offset(B, C, H)
            workspace[off:off + 256].zero_()
            carry = workspace[off:off + 8].view(torch.float64)
        if want_t:                              # torchdiffeq: aug_state[0] -= func(t[i], y[i]) . grad_y[i], all output times at once
            pre = torch.nn.functional.linear(torch.nn.functional.linear(z_saved[:, 1:], w1, b1).relu(), w2, b2)
            if self.act == _lib.ACT_TANH:
                pre = pre.tanh()
            dX = self.path.derivative(self.t_out[1:]).reshape(B, self.n_out - 1, C)
            terms = ((pre.view(B, self.n_out - 1, H, C) * dX.unsqueeze(2)).sum(-1) * grad_out[:, 1:]).sum((0, 2))
            if shared is not None:                # (as in run_adjoint: the time terms of the whole batch)
                terms64 = terms.to(torch.float64)
                shared[0](terms64)
                terms = terms64.to(torch.float32)
            for i in range(1, self.n_out):
                time_terms[i] = terms[i - 1]
        for i in range(self.n_out - 1, 0, -1):
            y = z_saved[:, i].contiguous()
            s0, s1 = -float(self.t_host[i]), -float(self.t_host[i - 1])
            if want_t:
                carry.copy_((carry.to(torch.float32) - time_terms[i]).to(torch.float64))
            launched = 0
            while True:
                if shared is None:
                    head = (_lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1), _lib.ptr(b1),
                            width, _lib.ptr(w2), _lib.ptr(b2), self.act, _lib.ptr(y), _lib.ptr(a), s0, s1, _lib.ptr(self.jump_s),
                            self.n_jump_s, self.adjoint_rtol, self.adjoint_atol, self.adj_safety, self.adj_ifactor,
                            self.adj_dfactor, self.adj_norm_kind, _lib.ptr(a_out), B, C, H, _lib.dtype_enum(torch.float32),
                            int(i == self.n_out - 1) | (2 if want_t else 0), _lib.ptr(workspace), workspace.numel(), launched,
                            _DOPRI_CHUNK)
                    if want_control:
                        _lib.check(lib.cde_dopri5_adjoint_mlp_advance_dcontrol(*head, _lib.ptr(grad_x), int(self.control_numel),
                                                                               _lib.ptr(grad_k), stream),
                                   "cde_dopri5_adjoint_mlp_advance_dcontrol")
                    else:
                        _lib.check(lib.cde_dopri5_adjoint_mlp_advance(*head, stream), "cde_dopri5_adjoint_mlp_advance")
                    launched += _DOPRI_CHUNK
                else:
                    # one controller for all shards (round 4): per attempted step ONE attempt launch, then this shard's
                    # state sums + S / E gradient images are all-reduced and every shard commits / measures the
                    # parameter blocks on the same (global) numbers
                    for _ in range(_DOPRI_CHUNK):
                        _lib.check(lib.cde_dopri5_adjoint_mlp_advance_sharded(
                            _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1),
                            _lib.ptr(b1), width, _lib.ptr(w2), _lib.ptr(b2), self.act, _lib.ptr(y), _lib.ptr(a), s0, s1,
                            _lib.ptr(self.jump_s), self.n_jump_s, self.adjoint_rtol, self.adjoint_atol, self.adj_safety,
                            self.adj_ifactor, self.adj_dfactor, self.adj_norm_kind, _lib.ptr(a_out), B, C, H,
                            _lib.dtype_enum(torch.float32), int(i == self.n_out - 1) | (2 if want_t else 0), _lib.ptr(workspace),
                            workspace.numel(), launched, _lib.ptr(reduced) if launched else None, shared[1], stream),
                            "cde_dopri5_adjoint_mlp_advance_sharded")
                        launched += 1
                        if state_only:
                            _lib.check(lib.cde_dopri5_adjoint_mlp_state_sums(_lib.ptr(workspace), workspace.numel(), B, C, H,
                                                                             launched, _lib.ptr(reduced), stream),
                                       "cde_dopri5_adjoint_mlp_state_sums")
                            shared[0](reduced)
                            _lib.check(lib.cde_dopri5_adjoint_mlp_apply_state_sums(
                                _lib.ptr(workspace), workspace.numel(), B, C, H, launched, _lib.ptr(reduced), stream),
                                "cde_dopri5_adjoint_mlp_apply_state_sums")
                            continue
                        _lib.check(lib.cde_dopri5_adjoint_mlp_pending_sums(_lib.ptr(workspace), workspace.numel(), B, C, H,
                                                                           launched, _lib.ptr(reduced), stream),
                                   "cde_dopri5_adjoint_mlp_pending_sums")
                        shared[0](reduced)
                        _lib.check(lib.cde_dopri5_adjoint_mlp_apply_reduced(
                            _lib.ptr(workspace), workspace.numel(), B, C, H, self.adjoint_rtol, self.adjoint_atol, launched,
                            _lib.ptr(reduced), stream), "cde_dopri5_adjoint_mlp_apply_reduced")
                at = (launched & 1) * stride
                status = _lib.DopriStatus.from_buffer_copy(workspace[at:at + size].cpu().numpy().tobytes())
                if status.phase == 4:
                    break
                if launched > 2_000_000:
                    raise RuntimeError("torchcde_amd: the dopri5 adjoint did not reach t = %g after %d attempted steps"
                                       % (-s1, launched))
            stats["n_accept"] += status.n_accept
            stats["n_reject"] += status.n_reject
            stats["launches"] += launched
            if self.record:
                off = lib.cde_dopri5_adjoint_mlp_trace_offset(B, C, H, 0)
                n = min(status.n_accept, 4096)
                steps.append(workspace[off:off + 24 * n].view(torch.float64).view(n, 3).cpu())
                off = lib.cde_dopri5_adjoint_mlp_trace_offset(B, C, H, 1)
                n = min(status.n_accept + status.n_reject, 16384)
                attempts.append(workspace[off:off + 40 * n].view(torch.float64).view(n, 5).cpu())
            a = a_out + grad_out[:, i - 1]
        off = lib.cde_dopri5_adjoint_mlp_gradient_offset(B, C, H)
        totals = workspace[off:off + 4 * (256 * 129 + 128 * 33)].view(torch.float32)
        acc2, acc1 = totals[:256 * 129].view(1, 256, 129), totals[256 * 129:].view(128, 33)
        off = lib.cde_dopri5_adjoint_mlp_gradient_upper_offset(B, C, H)
        if off:                                                  # 32 units x 16 channels: hidden units 16..31, a second image
            acc2 = torch.cat([acc2, workspace[off:off + 4 * 256 * 129].view(torch.float32).view(1, 256, 129)])
        grad_w2, grad_b2 = _output_layer_gradients(acc2, H, C, width)
        grad_w1 = acc1[:width, :H].contiguous()
        grad_b1 = acc1[:width, 32].contiguous()
        last_dopri5_adjoint_stats = self.owner.dopri5_adjoint
        last_dopri5_adjoint_stats.clear()
        last_dopri5_adjoint_stats.update(stats)
        if self.record:
            last_dopri5_adjoint_stats["steps"] = steps
            last_dopri5_adjoint_stats["attempts"] = attempts
        if want_t:
            time_terms[0] = carry.to(torch.float32).reshape(())
            return (a, grad_w1, grad_b1, grad_w2, grad_b2, torch.stack(time_terms)) + tail
        return (a, grad_w1, grad_b1, grad_w2, grad_b2) + tail

    def run(self, z0, weight, bias):
        lib = _lib.load()
        out = torch.empty(self.B, self.n_out, self.H, dtype=self.dtype, device=self.device)
        z0c = z0.detach().reshape(self.B, self.H).contiguous()
        if self.n_out == 1:
            out[:, 0] = z0c
            return out
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        dt = _lib.dtype_enum(self.dtype)
        nbytes = lib.cde_dopri5_workspace_bytes(self.B, self.C, self.H, dt)
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        workspace[:_WORKSPACE_HEAD].zero_()     # controller blocks + partial sums (a sharded run all-reduces them from launch 0)
        size = ctypes.sizeof(_lib.DopriStatus)
        launched = 0
        if self.hidden is not None:
            w1, b1 = self.hidden.weight.detach().contiguous(), self.hidden.bias.detach().contiguous()
        shared = self.shared
        if shared is not None:
            if self.hidden is not None:
                return self._run_shared(lib, shared, out, z0c, w, b, dt, workspace, w1, b1)
            return self._run_shared(lib, shared, out, z0c, w, b, dt, workspace)
        while True:
            if self.hidden is None:
                _lib.check(lib.cde_dopri5_advance(
                    _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w), _lib.ptr(b),
                    self.act, _lib.ptr(z0c), _lib.ptr(self.t_out), self.n_out, _lib.ptr(self.jump_t), self.n_jump,
                    self.rtol, self.atol, self.safety, self.ifactor, self.dfactor, _lib.ptr(out), self.B, self.C, self.H,
                    dt, self.variant, _lib.ptr(workspace), workspace.numel(), launched, _DOPRI_CHUNK,
                    _lib.stream_ptr(self.device)), "cde_dopri5_advance")
            else:
                _lib.check(lib.cde_dopri5_advance_mlp(
                    _lib.ptr(self.coeffs), _lib.ptr(self.knots), self.n_intervals, self.degree, _lib.ptr(w1), _lib.ptr(b1),
                    w1.size(0), _lib.ptr(w), _lib.ptr(b), self.act, _lib.ptr(z0c), _lib.ptr(self.t_out), self.n_out,
                    _lib.ptr(self.jump_t), self.n_jump, self.rtol, self.atol, self.safety, self.ifactor, self.dfactor,
                    _lib.ptr(out), self.B, self.C, self.H, dt, _lib.ptr(workspace), workspace.numel(), launched,
                    _DOPRI_CHUNK, _lib.stream_ptr(self.device)), "cde_dopri5_advance_mlp")
            launched += _DOPRI_CHUNK
            raw = workspace[(launched & 1) * size:(launched & 1) * size + size].cpu().numpy().tobytes()   # one sync per chunk
            status = _lib.DopriStatus.from_buffer_copy(raw)
            if status.phase == 4:
                break
            if launched > 2_000_000:
                raise RuntimeError("torchcde_amd: dopri5 did not reach t[-1] after %d attempted steps (t = %g, dt = %g)"
                                   % (launched, status.t_hi, status.dt))
        last_dopri5_stats = self.owner.dopri5            # (updated in place: callers may hold the dict itself)
        last_dopri5_stats.clear()
        last_dopri5_stats.update(n_accept=status.n_accept, n_reject=status.n_reject, launches=launched)
        if self.record:
            off = lib.cde_dopri5_trace_offset(self.B, self.C, self.H, dt)
            n = min(status.n_accept, 4096)            # CDE_DOPRI5_TRACE_STEPS
            last_dopri5_stats["steps"] = workspace[off:off + 24 * n].view(torch.float64).view(n, 3).cpu()
        return out


class _FusedDopri5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z0, weight, bias, plan, wants, t=None, knots=None, *control):
        # `control`: the path's buffers the control derivative reads, present (and differentiable inputs) only when the
        # coefficient tensor is among adjoint_params -- K4a then integrates dL/dcoeffs as one more block of the adjoint state
        # (`knots`: the path's knot times when they are in adjoint_params too: a fourth block)
        out = plan.run(z0, weight, bias)
        ctx.plan, ctx.wants = plan, wants
        ctx.t_like = t
        ctx.want_x = len(control) > 0
        ctx.want_knots = knots is not None
        ctx.knots_like = knots
        ctx.save_for_backward(out, weight, bias)
        return out.reshape(*plan.batch, plan.n_out, plan.H)

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        plan = ctx.plan
        out, weight, bias = ctx.saved_tensors
        want_t = ctx.t_like is not None and ctx.needs_input_grad[5]
        want_knots = ctx.want_knots and ctx.needs_input_grad[6]
        res = plan.run_adjoint(out, grad_out, weight, bias, want_t=want_t, want_control=ctx.want_x, want_knots=want_knots)
        grad_z0, grad_w, grad_b = res[:3]
        grad_t = None
        if want_t:                                # on `t`'s own device: cdeint accepts a CPU `t` next to GPU data
            grad_t = res[3].to(device=ctx.t_like.device, dtype=ctx.t_like.dtype)
        want_w, want_b = ctx.wants
        control_grads, grad_knots = (), None
        if ctx.want_x:
            grad_x, grad_knots = res[-2], res[-1]
            if grad_knots is not None:
                grad_knots = grad_knots.to(device=ctx.knots_like.device, dtype=ctx.knots_like.dtype)
                grad_knots = _with_fit_chain(plan, grad_knots, grad_x)
            control_grads = _control_gradients(plan, grad_x, True, False, ctx.needs_input_grad[7:], None)[1:]
        return (grad_z0.reshape(*plan.batch, plan.H) if ctx.needs_input_grad[0] else None,
                grad_w.view_as(weight) if (ctx.needs_input_grad[1] and want_w) else None,
                grad_b.view_as(bias) if (ctx.needs_input_grad[2] and want_b) else None, None, None, grad_t,
                grad_knots) + control_grads


class _FusedMlpDopri5(torch.autograd.Function):
    """The reference examples' training call with their own model: cdeint(X, CDEFunc, z0, X.interval) -- dopri5 forward
    (K4 with the two-layer field) and torchdiffeq's adaptive adjoint backward (K4am)."""

    @staticmethod
    def forward(ctx, z0, w1, b1, w2, b2, plan, t=None, knots=None, *control):
        # (`control` / `knots`: as in _FusedDopri5 -- the path's buffers / knot times when adjoint_params names them)
        out = plan.run(z0, w2, b2)
        ctx.plan = plan
        ctx.t_like = t
        ctx.want_x = len(control) > 0
        ctx.knots_like = knots
        ctx.save_for_backward(out, w1, b1, w2, b2)
        return out.reshape(*plan.batch, plan.n_out, plan.H)

    @staticmethod
    @_once_differentiable
    def backward(ctx, grad_out):
        out, w1, b1, w2, b2 = ctx.saved_tensors
        plan = ctx.plan
        need = ctx.needs_input_grad
        want_t = ctx.t_like is not None and need[6]
        want_knots = ctx.knots_like is not None and need[7]
        res = plan.run_adjoint_mlp(out, grad_out, w1, b1, w2, b2, want_t=want_t, want_control=ctx.want_x,
                                   want_knots=want_knots)
        grad_z0, gw1, gb1, gw2, gb2 = res[:5]
        grad_t = res[5].to(device=ctx.t_like.device, dtype=ctx.t_like.dtype) if want_t else None
        control_grads, grad_knots = (), None
        if ctx.want_x:
            grad_x, grad_knots = res[-2], res[-1]
            if grad_knots is not None:
                grad_knots = grad_knots.to(device=ctx.knots_like.device, dtype=ctx.knots_like.dtype)
                grad_knots = _with_fit_chain(plan, grad_knots, grad_x)
            control_grads = _control_gradients(plan, grad_x, True, False, need[8:], None)[1:]
        return (grad_z0.reshape(*plan.batch, plan.H) if need[0] else None, gw1 if need[1] else None,
                gb1 if need[2] else None, gw2 if need[3] else None, gb2 if need[4] else None, None, grad_t,
                grad_knots) + control_grads


# ------------------------------------------------------------------------------------------ front end
def _shape_errors(dX_shape, system_shape, z0):
    # messages of reference solver.py:7-33
    if tuple(dX_shape[:-1]) != tuple(z0.shape[:-1]):
        raise ValueError("X.derivative did not return a tensor with the same number of batch dimensions as z0. "
                         "X.derivative returned shape {} (meaning {} batch dimensions), whilst z0 has shape {} "
                         "(meaning {} batch dimensions)."
                         "".format(tuple(dX_shape), tuple(dX_shape[:-1]), tuple(z0.shape), tuple(z0.shape[:-1])))
    if tuple(system_shape[:-2]) != tuple(z0.shape[:-1]):
        raise ValueError("func did not return a tensor with the same number of batch dimensions as z0. func returned "
                         "shape {} (meaning {} batch dimensions), whilst z0 has shape {} (meaning {} batch"
                         " dimensions)."
                         "".format(tuple(system_shape), tuple(system_shape[:-2]), tuple(z0.shape),
                                   tuple(z0.shape[:-1])))
    if system_shape[-2] != z0.size(-1):
        raise ValueError("func did not return a tensor with the same number of hidden channels as z0. func returned "
                         "shape {} (meaning {} channels), whilst z0 has shape {} (meaning {} channels)."
                         "".format(tuple(system_shape), system_shape[-2], tuple(z0.shape), z0.size(-1)))
    if system_shape[-1] != dX_shape[-1]:
        raise ValueError("func did not return a tensor with the same number of input channels as X.derivative "
                         "returned. func returned shape {} (meaning {} channels), whilst X.derivative returned shape "
                         "{} (meaning {} channels)."
                         "".format(tuple(system_shape), system_shape[-1], tuple(dX_shape), dX_shape[-1]))


def _cdeint_tuple(X, func, z0, t, adjoint, backend, kwargs):
    """Tuple-valued state (reference solver.py:68-95): checks and messages as the reference, then the step-wise solver
    on the concatenated state (what torchdiffeq does with tuples); returns a tuple of (..., len(t), H_i) tensors."""
    if backend == "torchsde":
        raise NotImplementedError("torchcde_amd: the torchsde backend is outside the native hot path.")
    if backend != "torchdiffeq":
        raise ValueError(f"Unrecognised backend={backend}")
    for part in z0:
        if not isinstance(part, torch.Tensor):
            raise ValueError("z0 must either a tensor or a tuple/list of tensors.")
        _lib.require_gpu(part, "z0")
    probe_t = t[0].to(z0[0].device) if isinstance(t, torch.Tensor) else t
    with torch.no_grad():
        dX = X.derivative(probe_t)
        if not isinstance(dX, (tuple, list)):
            raise ValueError("z0 is a tuple/list and so X.derivative must return a tuple/list as well.")
        if len(z0) != len(dX):
            raise ValueError("z0 and X.derivative(t) must be tuples of the same length.")
        is_prod = hasattr(func, "prod")
        system = func.prod(probe_t, z0, dX) if is_prod else func(probe_t, z0)
    name = "func.prod" if is_prod else "func"
    if not isinstance(system, (tuple, list)):
        raise ValueError("z0 is a tuple/list and so %s must return a tuple/list as well." % name)
    if len(z0) != len(system):
        raise ValueError("z0 and %s must be tuples of the same length."
                         % ("func.prod(t, z, dXdt)" if is_prod else "func(t, z)"))
    for dX_i, system_i, z_i in zip(dX, system, z0):
        if not isinstance(dX_i, torch.Tensor):
            raise ValueError("X.derivative must return a tensor or tuple of tensors.")
        if not isinstance(system_i, torch.Tensor):
            raise ValueError("%s must return a tensor or tuple/list of tensors." % name)
        if is_prod:
            if tuple(dX_i.shape[:-1]) != tuple(z_i.shape[:-1]):
                _shape_errors(tuple(dX_i.shape), tuple(z_i.shape) + (dX_i.size(-1),), z_i)
            if system_i.shape != z_i.shape:
                raise ValueError("func.prod did not return a tensor with the same shape as z0. func.prod returned shape "
                                 "{} whilst z0 has shape {}.".format(tuple(system_i.shape), tuple(z_i.shape)))
        else:
            _shape_errors(tuple(dX_i.shape), tuple(system_i.shape), z_i)
    from . import stepwise
    kw = dict(kwargs)
    kw.pop("variant", None)
    kw.setdefault("atol", 1e-6)
    kw.setdefault("rtol", 1e-4)
    if adjoint:
        kw.setdefault("adjoint_atol", kw["atol"])
        kw.setdefault("adjoint_rtol", kw["rtol"])
    sizes = [part.size(-1) for part in z0]
    field = stepwise.TupleField(X, func, sizes)
    out = stepwise.solve(X, func, torch.cat(z0, dim=-1), t, adjoint, kw.get("method") or "dopri5", kw.get("options"),
                         kw["rtol"], kw["atol"], kw.get("adjoint_method"), kw.get("adjoint_options"),
                         kw.get("adjoint_rtol"), kw.get("adjoint_atol"), kw.get("adjoint_params"), field=field)
    return tuple(out.split(sizes, dim=-1))


def _cdeint_foreign_control(X, func, z0, t, adjoint, kwargs):
    """A control that is not one of this package's paths (reference solver.py:45-46: anything with a `derivative` method,
    typically an nn.Module of the user's): the reference's compatibility checks and messages (solver.py:44-67), then the
    step-wise solver -- X.derivative(t) is called as it is (under autograd when its parameters or the times need a
    gradient), the vector field times dX/dt runs in cde_contract."""
    _lib.require_gpu(z0, "z0")
    probe_t = t[0].to(z0.device) if isinstance(t, torch.Tensor) else t
    is_prod = hasattr(func, "prod")
    with torch.no_grad():
        dX = X.derivative(probe_t.detach() if isinstance(probe_t, torch.Tensor) else probe_t)
        system = func.prod(probe_t, z0, dX) if is_prod else func(probe_t, z0)
    if not isinstance(dX, torch.Tensor):
        raise ValueError("z0 is a tensor and so X.derivative must return a tensor as well.")
    if is_prod:
        if not isinstance(system, torch.Tensor):
            raise ValueError("z0 is a tensor and so func.prod must return a tensor as well.")
        if tuple(dX.shape[:-1]) != tuple(z0.shape[:-1]):
            _shape_errors(tuple(dX.shape), tuple(z0.shape) + (dX.size(-1),), z0)
        if system.shape != z0.shape:
            raise ValueError("func.prod did not return a tensor with the same shape as z0. func.prod returned shape {} "
                             "whilst z0 has shape {}.".format(tuple(system.shape), tuple(z0.shape)))
    else:
        if not isinstance(system, torch.Tensor):
            raise ValueError("z0 is a tensor and so func must return a tensor as well.")
        _shape_errors(tuple(dX.shape), tuple(system.shape), z0)
    _lib.require_gpu(dX, "X.derivative(t)")
    buffers = tuple(X.buffers()) if isinstance(X, torch.nn.Module) else ()
    if adjoint and "adjoint_params" not in kwargs:
        for buffer in buffers:
            if buffer.requires_grad:
                warnings.warn(_GRAD_WARNING)
    from . import stepwise
    dispatch.record(dispatch.Choice(dispatch.STEPWISE, "the control is not one of this package's paths: X.derivative is "
                                                       "called at every evaluation"), None)
    kw = dict(kwargs)
    field = stepwise.ForeignControlField(X, func)
    return stepwise.solve(X, func, z0, t, adjoint, kw.pop("method", None) or "dopri5", kw.pop("options", None),
                          kw["rtol"], kw["atol"], kw.get("adjoint_method"), kw.get("adjoint_options"),
                          kw.get("adjoint_rtol"), kw.get("adjoint_atol"), kw.get("adjoint_params"), field=field)


def cdeint(X, func, z0, t, adjoint=True, backend="torchdiffeq", **kwargs):
    r"""Solve  z_t = z_{t_0} + \int_{t_0}^t f(s, z_s) dX_s  on the MI355X.

    Arguments, return value (shape ``(..., len(t), hidden_channels)``) and errors as reference
    ``torchcde.cdeint`` (solver.py:144-194).  ``variant=`` (extra keyword, one of
    "auto" | "generic" | "mfma" | "split" | "bf16x3") selects the kernel family: "bf16x3" runs the rk4 solve's weight
    GEMMs on the bf16 matrix pipe at float32 accuracy (csrc/rk4_bf16x3.hip), the others exist for testing."""
    variant = {"auto": _lib.VARIANT_AUTO, "generic": _lib.VARIANT_GENERIC, "mfma": _lib.VARIANT_MFMA,
               "split": _lib.VARIANT_SPLIT, "bf16x3": _lib.VARIANT_BF16X3}[kwargs.pop("variant", "auto")]
    # tolerance defaults of solver.py:195-203 (only adaptive methods read them)
    kwargs.setdefault("atol", 1e-6)
    kwargs.setdefault("rtol", 1e-4)
    if adjoint:
        kwargs.setdefault("adjoint_atol", kwargs["atol"])
        kwargs.setdefault("adjoint_rtol", kwargs["rtol"])

    if not hasattr(X, "derivative"):
        raise ValueError("X must have a 'derivative' method.")
    if isinstance(z0, (tuple, list)):
        return _cdeint_tuple(X, func, tuple(z0), t, adjoint, backend, kwargs)
    if not isinstance(z0, torch.Tensor):
        raise ValueError("z0 must either a tensor or a tuple/list of tensors.")
    if backend == "torchsde":
        raise NotImplementedError("torchcde_amd: the torchsde backend is outside the native hot path.")
    if backend != "torchdiffeq":
        raise ValueError(f"Unrecognised backend={backend}")
    if not isinstance(X, _NativePath):
        # solver.py:45-46 asks X for nothing but a `derivative` method: a user-defined control is solved step by step on the
        # GPU with X.derivative(t) itself under every evaluation (cde_contract for the matrix-vector product)
        return _cdeint_foreign_control(X, func, z0, t, adjoint, kwargs)
    stepwise_kwargs = dict(kwargs)      # what the step-wise path would receive (the reference forwards these verbatim)
    _lib.require_gpu(z0, "z0")
    packed = X._packed()
    _lib.require_gpu(packed, "the control path")
    batch = tuple(packed.shape[:-2])
    C = X._channels()
    H = z0.size(-1)
    if hasattr(func, "prod"):
        # solver.py:48-53, :35-41, :121-123: the module computes f(t, z) dX itself; solved step by step with the native
        # control derivative under every evaluation
        if batch != tuple(z0.shape[:-1]):
            _shape_errors(batch + (C,), tuple(z0.shape) + (C,), z0)
        with torch.no_grad():
            probe_t = t[0].to(z0.device) if isinstance(t, torch.Tensor) else t
            first = func.prod(probe_t, z0, X.derivative(probe_t))
        if not isinstance(first, torch.Tensor):
            raise ValueError("z0 is a tensor and so func.prod must return a tensor as well.")
        if first.shape != z0.shape:
            raise ValueError("func.prod did not return a tensor with the same shape as z0. func.prod returned shape {} "
                             "whilst z0 has shape {}.".format(tuple(first.shape), tuple(z0.shape)))
        from . import stepwise
        dispatch.record(dispatch.Choice(dispatch.STEPWISE, "func.prod multiplies by the control derivative itself"), None)
        kw = stepwise_kwargs
        return stepwise.solve(X, func, z0, t, adjoint, kw.pop("method", None) or "dopri5", kw.pop("options", None),
                              kw["rtol"], kw["atol"], kw.get("adjoint_method"), kw.get("adjoint_options"),
                              kw.get("adjoint_rtol"), kw.get("adjoint_atol"), kw.get("adjoint_params"))

    # compatibility probe of solver.py:44-67: one evaluation of func at t[0], shapes checked against z0;
    # the same evaluation establishes (bitwise) whether func belongs to the fused affine family.
    field, system = probe(func, t[0].to(z0.device) if isinstance(t, torch.Tensor) else t, z0)
    if not isinstance(system, torch.Tensor):
        raise ValueError("z0 is a tensor and so func must return a tensor as well.")
    _shape_errors(batch + (C,), tuple(system.shape), z0)
    # func's formula, when the probe identified it: lets the step-wise adjoint write its dynamics in closed form
    # (`variant="generic"` keeps autograd: the independent cross-check the tests compare against)
    recognised = field if variant == _lib.VARIANT_AUTO else None
    recognised_kind = None if field is None else field.kind
    mlp = None
    if field is not None and field.kind == "mlp2":
        mlp, field = (field if _mlp_fusable(field, H, C, z0, packed) else None), None
    if field is not None:
        weight, bias = field.weight, field.bias
        if tuple(weight.shape) != (H * C, H) or not (z0.dtype == packed.dtype == weight.dtype):
            field = None              # not a shape / dtype the fused kernels take: solve it step by step instead
        elif not _lib.load().cde_rk4_supported(C, H, _lib.dtype_enum(z0.dtype), field.act, int(bool(adjoint)), variant):
            field = None              # beyond the fused kernels' tiles (LDS / lane limits): step-wise, not a late error

    if variant == _lib.VARIANT_BF16X3 and (field is None or field.act != _lib.ACT_NONE or kwargs.get("method") != "rk4"
                                           or z0.dtype != torch.float32 or H > 32 or C > 8):
        # an explicit request for a specific kernel family is never redirected
        raise NotImplementedError("torchcde_amd: variant='bf16x3' covers method='rk4' with the one-layer identity-activation "
                                  "field Linear(H, H*C), float32, H <= 32, C <= 8.")
    if adjoint and "adjoint_params" not in kwargs:
        for buffer in X.buffers():
            if buffer.requires_grad:
                warnings.warn(_GRAD_WARNING)

    # solver configuration (what the reference forwards verbatim to torchdiffeq, solver.py:175-176,227)
    method = kwargs.pop("method", None)
    options = kwargs.pop("options", None)
    if method is None:
        method = "dopri5"
    grad_mode = torch.is_grad_enabled()
    given_params = kwargs.get("adjoint_params")
    if given_params is not None:
        given_params = tuple(given_params)
    # what must be differentiated: z0, the field's parameters (or the caller's adjoint_params -- which may name the
    # control's coefficient / knot tensors, README.md:251-270), the output times
    param_source = given_params if (adjoint and given_params is not None) else (
        tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ())
    wants_grad = grad_mode and (z0.requires_grad or any(isinstance(p, torch.Tensor) and p.requires_grad
                                                        for p in param_source))
    if not adjoint and grad_mode and any(b.requires_grad for b in X.buffers()):
        wants_grad = True            # backprop through the solver reaches the control: the step-wise path differentiates it
    wants_t = grad_mode and isinstance(t, torch.Tensor) and t.requires_grad
    wants_grad = wants_grad or wants_t
    # gradients w.r.t. the control's tensors or the times come out of the pre-activation MFMA adjoint kernel only
    control_ids = {b.untyped_storage().data_ptr() for b in X._control_buffers()} | {X._t.untyped_storage().data_ptr()}
    control_wants = grad_mode and adjoint and given_params is not None and any(
        isinstance(p, torch.Tensor) and p.requires_grad and p.untyped_storage().data_ptr() in control_ids
        for p in given_params)
    if not adjoint:
        # autograd through the solver's own operations reaches every control tensor that requires a gradient
        control_wants = grad_mode and any(b.requires_grad for b in X.buffers())
    mfma_shape = (field is not None and z0.dtype == torch.float32 and H <= 32 and C <= 8
                  and variant != _lib.VARIANT_GENERIC)
    adj_opts = kwargs.get("adjoint_options")
    adjoint_method = kwargs.get("adjoint_method")
    t_is_vector = isinstance(t, torch.Tensor) and t.dim() == 1 and t.is_floating_point() and t.numel() >= 1
    t_host = _to_host(t) if t_is_vector else None
    increasing = t_is_vector and (t_host.numel() == 1 or bool((t_host[1:] > t_host[:-1]).all()))

    # ---- how the caller's adjoint_params relate to the field's own parameters and the control's tensors
    known = field if field is not None else mlp
    own = () if known is None else (
        (field.weight, field.bias) if field is not None else
        (mlp.hidden.weight, mlp.hidden.bias, mlp.output.weight, mlp.output.bias))
    params_kind = "default"
    if given_params is not None and known is not None:
        extra = [p for p in given_params if not any(p is o for o in own)]
        extra_ok = all(isinstance(p, torch.Tensor) and p.untyped_storage().data_ptr() in control_ids for p in extra)
        complete = field is not None or all(any(p is o for p in given_params) for o in own)    # K3m / K4am: all four or none
        params_kind = "own" if (extra_ok and complete) else "foreign"
        if field is None and any(p is X._t for p in extra) and method not in ("rk4", "dopri5"):
            params_kind = "foreign"             # knot-time gradients of a two-layer solve: under rk4 and (with the coefficient
                                                # tensor: control_block) dopri5, else step-wise

    # the adaptive backward measures every entry of adjoint_params as a block of its error norm: control gradients are fused
    # there when the ONE extra entry is the packed coefficient tensor the path reads (its buffers are views of it)
    control_block = None
    if given_params is not None and known is not None and params_kind == "own":
        extra = [p for p in given_params if not any(p is o for o in own)]
        tensors = [p for p in extra if p is not X._t]
        if (len(tensors) == 1 and len(extra) <= 2 and tensors[0].requires_grad and tensors[0].is_contiguous()
                and tensors[0].numel() == packed.numel() and tensors[0].dtype == packed.dtype
                and tensors[0].untyped_storage().data_ptr() == packed.untyped_storage().data_ptr()):
            control_block = tensors[0]          # (+ optionally the knot times X._t: a fourth block)

    fixed_keys, adaptive_keys = {"step_size"}, {"jump_t", "safety", "ifactor", "dfactor"}
    # ONE normalised view of the options for the fused paths (the step-wise path gets them verbatim, like torchdiffeq):
    # None values and torchdiffeq's own defaults spelled out are the same request as leaving the key away
    fused_options = _strip_noops(options, fixed=method != "dopri5")
    fused_adj_opts = _strip_noops(adj_opts, fixed=method != "dopri5")

    def within(opts, allowed):
        return opts is None or (isinstance(opts, dict) and set(opts) <= allowed)

    if method == "dopri5":
        options_ok = within(fused_options, adaptive_keys)
        adjoint_options_ok = fused_adj_opts is None or (within(fused_adj_opts, adaptive_keys | {"norm"})
                                                        and fused_adj_opts.get("norm", "seminorm") == "seminorm")
    else:
        options_ok = within(fused_options, fixed_keys)
        adjoint_options_ok = within(fused_adj_opts, fixed_keys)
    from .distributed import step_control
    request = dispatch.Request(
        prod=False, kind=None if known is None else known.kind, tiles_ok=known is not None, mfma_shape=mfma_shape,
        method=method, adjoint=bool(adjoint), wants_grad=bool(wants_grad), wants_t=bool(wants_t),
        wants_control=bool(control_wants), params=params_kind,
        adjoint_method_ok=adjoint_method in (None, method), options_ok=options_ok,
        adjoint_options_ok=adjoint_options_ok, t_ok=bool(increasing) or not t_is_vector,
        variant_generic=variant == _lib.VARIANT_GENERIC, shared=step_control() is not None, narrow_control=C <= 8,
        backprop_ok=bool((mfma_shape and variant in (_lib.VARIANT_AUTO, _lib.VARIANT_MFMA))
                         or (mlp is not None and variant == _lib.VARIANT_AUTO)),
        identity=bool(field is not None and field.act == _lib.ACT_NONE),
        control_block=control_block is not None)
    if recognised_kind is not None and known is None:
        # the probe recognised the formula but the shape / dtype is beyond the tiles: say so in the record and the warning
        request = request._replace(kind=recognised_kind, tiles_ok=False)
    choice = dispatch.select_path(request)
    if (mlp is not None and _mlp_upper_half(H, C) and choice.path != dispatch.STEPWISE
            and (choice.path not in _UPPER_HALF_PATHS or (choice.path == "mlp_dopri5_adjoint" and request.shared))):
        # 32 units x 16 channels: fused where the kernels read the upper half (see _UPPER_HALF_PATHS)
        request = request._replace(tiles_ok=False)
        choice = dispatch.select_path(request)
    dispatch.record(choice, request)

    if choice.path == dispatch.STEPWISE:
        # Arbitrary vector fields / methods / differentiation modes: host-driven stepping with the native control
        # derivative and contraction kernels under every evaluation (torchcde_amd/stepwise.py).
        from . import stepwise
        dispatch.warn_once(func, choice, request)
        if step_control() is not None and (method == "dopri5" or adjoint_method == "dopri5"):
            warnings.warn("torchcde_amd: shared_step_control() is active but this adaptive solve runs step-wise, which "
                          "does not share its step controller across ranks: every rank takes its own step sequence.")
        kw = stepwise_kwargs
        return stepwise.solve(X, func, z0, t, adjoint, method, options, kw["rtol"], kw["atol"],
                              kw.get("adjoint_method"), kw.get("adjoint_options"), kw.get("adjoint_rtol"),
                              kw.get("adjoint_atol"), kw.get("adjoint_params"), recognised=recognised)
    if not t_is_vector:
        raise ValueError("t must be a one dimensional floating point tensor.")

    unknown = set(kwargs) - {"rtol", "atol", "adjoint_rtol", "adjoint_atol", "adjoint_method", "adjoint_options",
                             "adjoint_params"}
    if unknown:
        raise NotImplementedError("torchcde_amd: unsupported cdeint keyword arguments {}".format(sorted(unknown)))

    # ---- two-layer fields
    if choice.path == "mlp_rk4_adjoint":
        step = _parse_fixed_options(fused_options, "solver")
        adj_step = step if fused_adj_opts is None else _parse_fixed_options(fused_adj_opts, "adjoint")
        plan = _MlpPlan(X, mlp, batch, H, C, t, step, adj_step)
        want_knots = bool(grad_mode and given_params is not None and any(p is X._t and p.requires_grad for p in given_params))
        # (the coefficient tensor the path was built from: dL/dcoeffs flows back through the path's buffer views)
        want_x = bool(grad_mode and given_params is not None and any(
            isinstance(p, torch.Tensor) and p.requires_grad and p is not X._t
            and p.untyped_storage().data_ptr() in control_ids for p in given_params))
        control_inputs = X._control_buffers() if want_x else ()
        plan.fit_chain = _knot_fit_chain(X) if want_knots else None
        return _FusedMlpRK4.apply(z0, mlp.hidden.weight, mlp.hidden.bias, mlp.output.weight, mlp.output.bias, plan,
                                  want_x, t if wants_t else None, X._t if want_knots else None, *control_inputs)
    if choice.path in ("rk4_backprop", "mlp_rk4_backprop"):
        # adjoint=False: the control tensors autograd must reach (their gradients come out of the same reverse-mode sweep)
        want_x = bool(grad_mode and any(b.requires_grad for b in X._control_buffers()))
        knots_in = X._t if (grad_mode and X._t.requires_grad) else None
        control_inputs = X._control_buffers() if want_x else ()
    if choice.path == "mlp_rk4_backprop":
        plan = _MlpPlan(X, mlp, batch, H, C, t, _parse_fixed_options(fused_options, "solver"))
        return _FusedMlpRK4Backprop.apply(z0, mlp.hidden.weight, mlp.hidden.bias, mlp.output.weight, mlp.output.bias, plan,
                                          want_x, knots_in, *control_inputs)
    if choice.path == "mlp_rk4_forward":
        with torch.no_grad():
            return _MlpPlan(X, mlp, batch, H, C, t, _parse_fixed_options(fused_options, "solver")).run(z0)
    if choice.path in ("mlp_dopri5_forward", "mlp_dopri5_adjoint"):
        plan = _Dopri5Plan(X, mlp, batch, H, C, t, kwargs["rtol"], kwargs["atol"], fused_options, variant,
                           kwargs.get("adjoint_rtol"), kwargs.get("adjoint_atol"), fused_adj_opts)
        if choice.path == "mlp_dopri5_forward":
            with torch.no_grad():
                return plan.run(z0, mlp.weight, mlp.bias).reshape(*batch, plan.n_out, H)
        # the reference examples' own training call (no method: dopri5 + adjoint): K4 forward, K4am backward
        want_x = bool(control_wants and control_block is not None)
        control_inputs = X._control_buffers() if want_x else ()
        plan.control_numel = control_block.numel() if want_x else 0
        knots_in = X._t if (want_x and any(p is X._t and p.requires_grad for p in given_params)) else None
        plan.fit_chain = _knot_fit_chain(X) if knots_in is not None else None
        return _FusedMlpDopri5.apply(z0, mlp.hidden.weight, mlp.hidden.bias, mlp.output.weight, mlp.output.bias, plan,
                                     t if wants_t else None, knots_in, *control_inputs)

    # ---- one-layer fields
    weight, bias = field.weight, field.bias
    if choice.path == "rk4_backprop":
        plan = _Plan(X, field, batch, H, C, t, _parse_fixed_options(fused_options, "solver"), None, False, variant)
        return _FusedRK4Backprop.apply(z0, weight, bias, plan, want_x, knots_in, *control_inputs)
    if choice.path in ("dopri5_forward", "dopri5_adjoint"):
        plan = _Dopri5Plan(X, field, batch, H, C, t, kwargs["rtol"], kwargs["atol"], fused_options, variant,
                           kwargs.get("adjoint_rtol"), kwargs.get("adjoint_atol"), fused_adj_opts)
        if choice.path == "dopri5_forward":
            # nothing is to be differentiated (`adjoint_params=()` with weights that still require grad, no_grad ...): no
            # autograd node -- one here would reach K4a without its eligibility checks
            with torch.no_grad():
                return plan.run(z0, weight, bias).reshape(*batch, plan.n_out, H)
        wants = (True, True) if given_params is None else (any(p is weight for p in given_params),
                                                           any(p is bias for p in given_params))
        # control gradients: dL/dcoeffs flows back to the caller's tensor through the path's buffer views
        want_x = bool(control_wants and control_block is not None)
        control_inputs = X._control_buffers() if want_x else ()
        plan.control_numel = control_block.numel() if want_x else 0
        knots_in = X._t if (want_x and any(p is X._t and p.requires_grad for p in given_params)) else None
        plan.fit_chain = _knot_fit_chain(X) if knots_in is not None else None
        return _FusedDopri5.apply(z0, weight, bias, plan, wants, t if wants_t else None, knots_in, *control_inputs)
    step_size = _parse_fixed_options(fused_options, "solver")
    adjoint_step = step_size if fused_adj_opts is None else _parse_fixed_options(fused_adj_opts, "adjoint")
    want_w = want_b = True
    want_x = want_knots = False
    if given_params is not None and adjoint:
        for p in given_params:
            if p is weight or p is bias:
                continue
            if p is X._t:
                # the knot times of the control (reference test/test_tricks.py:21-49 passes them): the chain through
                # `frac = t - t_j` of the spline evaluation
                want_knots = want_knots or (p.requires_grad and grad_mode)
            else:
                # the coefficient tensor the path was built from (README.md:251-270): dL/dcoeffs flows back to it
                # through the path's buffer views
                want_x = want_x or (p.requires_grad and grad_mode)
        want_w = any(p is weight for p in given_params)
        want_b = any(p is bias for p in given_params)
    plan = _Plan(X, field, batch, H, C, t, step_size, adjoint_step, adjoint, variant,
                 _lib.FIXED_METHODS[method] if choice.path == "fixed_grid" else _lib.METHOD_RK4)
    control_inputs = X._control_buffers() if want_x else ()
    plan.fit_chain = _knot_fit_chain(X) if want_knots else None
    return _FusedRK4.apply(z0, weight, bias, plan, (want_w, want_b, want_x), t if wants_t else None,
                           X._t if want_knots else None, *control_inputs)


# ------------------------------------------------------------------------------------------ per-thread module attributes
class _Module(types.ModuleType):
    """`torchcde_amd.cdeint.last_dopri5_stats` etc. read and write the CALLING THREAD's state (_CallState): two threads
    solving at once do not see each other's statistics, and one thread recording step traces does not switch it on for
    the other."""
    last_dopri5_stats = property(lambda self: _state().dopri5)
    last_dopri5_adjoint_stats = property(lambda self: _state().dopri5_adjoint)
    record_dopri5_steps = property(lambda self: _state().record, lambda self, value: setattr(_state(), "record", bool(value)))
    event_log = property(lambda self: _state().event_log, lambda self, value: setattr(_state(), "event_log", value))


sys.modules[__name__].__class__ = _Module
