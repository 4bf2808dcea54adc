"""The log-ODE transform of the reference (torchcde/log_ode.py:15-133) on the GPU.

Host side: the window bookkeeping of log_ode.py:18-49 (a loop over the handful of window end times, on a host copy
of ``t``), the merge of the new times into the series (torch indexing on the device) and the linear fill
(``linear_interpolation_coeffs`` -> K0).  Device side (K5, ``cde_logsig_windows``): the logsignature of every window
and their running sum -- the arithmetic the reference obtains from the ``signatory`` package, in its default "words"
basis (coefficients of the Lyndon words in the expanded logsignature).
"""
import torch

from . import _lib
from .paths import _validate_input_path, _no_grad_through_path, linear_interpolation_coeffs


def _lyndon_words(channels, depth):
    """Lyndon words over {0..channels-1} up to length ``depth``, ordered by length then lexicographically (Duval's
    generation algorithm) -- the coordinate order of signatory's logsignatures."""
    found, word = [], [-1]
    while word:
        word[-1] += 1
        found.append(tuple(word))
        period = len(word)
        while len(word) < depth:
            word.append(word[len(word) - period])
        while word and word[-1] == channels - 1:
            word.pop()
    return sorted((w for w in found if len(w) <= depth), key=lambda w: (len(w), w))


def logsignature_channels(channels, depth):
    return len(_lyndon_words(channels, depth))


_plans = {}          # window bookkeeping per (times, window length, shape): host work done once, device tables reused


def _plan(t, given, window_length, L, C_in, depth, version, dtype, device):
    """log_ode.py:18-49 for one set of times: which rows of the (merged) series bound the windows, which new times have
    to be merged in, the per-window scale and the Lyndon-word table -- as device tensors.  The reference's loop
    (`value <= t[pointer]` or `value.allclose(t[pointer])`, pointer never moving back) is evaluated for all window ends
    at once with the same elementwise arithmetic (`torch.isclose` is allclose's formula)."""
    from .cdeint import _to_host          # id + version guarded host copy: steady-state calls do not synchronise
    th = _to_host(t)
    # keyed on the VALUES of the times (as cdeint._grids_for does), never on an address: allocators recycle blocks, and a
    # different `t` of the same length at a recycled address must not be handed the previous plan
    key = ((th.numpy().tobytes(), str(th.dtype)) if given else None, float(window_length), L, C_in, depth, version, dtype,
           str(device))
    plan = _plans.get(key)
    if plan is not None:
        return plan
    timespan = th[-1] - th[0]
    pieces = int((timespan / window_length).ceil().item())
    new_t = torch.linspace(th[0].item(), (th[0] + pieces * window_length).item(), pieces + 1, dtype=th.dtype)
    new_t = torch.min(new_t, th.max())
    close = torch.isclose(new_t[:, None], th[None, :])
    hit = (new_t[:, None] <= th[None, :]) | close
    pointer = hit.to(torch.int8).argmax(dim=1)                   # first time at or after (or close to) every window end
    is_close = close[torch.arange(new_t.numel()), pointer]
    n_fresh_before = torch.cumsum((~is_close).to(torch.int64), 0) - (~is_close).to(torch.int64)
    rows = (pointer + n_fresh_before).tolist()
    fresh = new_t[~is_close]
    order_dev = merged_dev = None
    if fresh.numel():                                           # the new times join the series as missing observations
        merged, order = torch.cat([th, fresh]).sort()
        order_dev, merged_dev = order.clamp(0, L).to(device), merged.to(device)
    n_windows = len(rows) - 1
    scale = (new_t[1:] - new_t[:-1]) if version == 0 else torch.ones(n_windows, dtype=th.dtype)
    table = []
    for word in _lyndon_words(C_in, depth):
        flat = 0
        for letter in word:
            flat = flat * C_in + letter
        table.append((len(word), flat))
    plan = dict(rows=torch.tensor(rows, dtype=torch.int64).to(device), order=order_dev, merged=merged_dev,
                scale=scale.to(device=device, dtype=dtype).contiguous(),
                words=torch.tensor(table, dtype=torch.int32).to(device), n_windows=n_windows, n_words=len(table),
                new_t=new_t.to(device))
    if len(_plans) > 64:
        _plans.clear()
    _plans[key] = plan
    return plan


class _LogsigWindows(torch.autograd.Function):
    """K5 with its backward (``cde_logsig_windows_backward``): signatory's logsignatures are differentiable w.r.t. the
    path, so the reference's ``logsig_windows`` is too (log_ode.py:53-63)."""

    @staticmethod
    def forward(ctx, x, rows, scale, words, depth, n_windows, n_words):
        L, C = x.size(-2), x.size(-1)
        src = x.detach().contiguous()
        B = src.numel() // (L * C)
        out = torch.empty(*x.shape[:-2], n_windows + 1, n_words, dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().cde_logsig_windows(_lib.ptr(src), _lib.ptr(rows), _lib.ptr(scale), _lib.ptr(words),
                                                  _lib.ptr(out), B, L, C, depth, n_windows, n_words,
                                                  _lib.dtype_enum(x.dtype), _lib.stream_ptr(x.device)),
                   "cde_logsig_windows")
        ctx.save_for_backward(src, rows, scale, words)
        ctx.meta = (depth, n_windows, n_words)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        src, rows, scale, words = ctx.saved_tensors
        depth, n_windows, n_words = ctx.meta
        L, C = src.size(-2), src.size(-1)
        B = src.numel() // (L * C)
        grad_out = grad_out.contiguous()
        grad_x = torch.empty_like(src)
        workspace = torch.empty_like(grad_out)
        _lib.check(_lib.load().cde_logsig_windows_backward(
            _lib.ptr(grad_out), _lib.ptr(src), _lib.ptr(rows), _lib.ptr(scale), _lib.ptr(words), _lib.ptr(grad_x),
            _lib.ptr(workspace), B, L, C, depth, n_windows, n_words, _lib.dtype_enum(src.dtype),
            _lib.stream_ptr(src.device)), "cde_logsig_windows_backward")
        return grad_x, None, None, None, None, None, None


def _windows(x, depth, window_length, t, version):
    given = t is not None
    t = _validate_input_path(x, t)
    _lib.require_gpu(x, "x")
    _no_grad_through_path(t)
    C_in = x.size(-1)
    if not ((1 <= depth <= 3 and C_in <= 8) or (depth == 4 and C_in <= 5) or (1 <= depth <= 2 and C_in <= 32)):
        raise NotImplementedError("torchcde_amd: logsignatures are implemented natively for depth <= 3 with at most 8 "
                                  "channels, depth 4 with at most 5 and depth <= 2 with at most 32 (got depth=%d, "
                                  "channels=%d)." % (depth, C_in))
    plan = _plan(t, given, window_length, x.size(-2), C_in, depth, version, x.dtype, x.device)
    batch = x.shape[:-2]
    t_dev = t.to(x.device)
    if plan["order"] is not None:                              # merge the new times in as missing observations
        missing = torch.full((*batch, 1, x.size(-1)), float("nan"), dtype=x.dtype, device=x.device)
        x = torch.cat([x, missing], dim=-2)[..., plan["order"], :]
        t_dev = plan["merged"]
    x = linear_interpolation_coeffs(x, t_dev)                  # fills the NaNs (the new rows and any in the data)
    out = _LogsigWindows.apply(x, plan["rows"], plan["scale"], plan["words"], depth, plan["n_windows"], plan["n_words"])
    if version == 0:
        return out, plan["new_t"].clone()                    # the cached tensor itself is never handed out
    return out


def logsig_windows(x, depth, window_length, t=None):
    """Logsignatures over windows of length ``window_length``, accumulated (reference log_ode.py:107-133): the values of
    the transformed path at times 0, 1, 2, ...; feed them to ``linear_interpolation_coeffs`` / ``LinearInterpolation``."""
    return _windows(x, depth, window_length, t, 1)


def logsignature_windows(x, depth, window_length, t=None):
    """Deprecated variant of the reference (log_ode.py:78-104): every window's logsignature is scaled by the window
    length and the window times are returned as well."""
    return _windows(x, depth, window_length, t, 0)
