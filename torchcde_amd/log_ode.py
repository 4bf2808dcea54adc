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


def _windows(x, depth, window_length, t, version):
    t = _validate_input_path(x, t)
    _lib.require_gpu(x, "x")
    _no_grad_through_path(x, t)
    C_in = x.size(-1)
    if not ((1 <= depth <= 3 and C_in <= 8) or (depth == 4 and C_in <= 5) or (1 <= depth <= 2 and C_in <= 32)):
        raise NotImplementedError("torchcde_amd: logsignatures are implemented natively for depth <= 3 with at most 8 "
                                  "channels, depth 4 with at most 5 and depth <= 2 with at most 32 (got depth=%d, "
                                  "channels=%d)." % (depth, C_in))
    words = _lyndon_words(C_in, depth)
    th = t.detach().cpu()
    # log_ode.py:18-40 on the host copy of the times
    timespan = th[-1] - th[0]
    pieces = int((timespan / window_length).ceil().item())
    new_t = torch.linspace(th[0].item(), (th[0] + pieces * window_length).item(), pieces + 1, dtype=th.dtype)
    new_t = torch.min(new_t, th.max())
    pointer, fresh, rows = 0, [], []
    for value in new_t:
        while True:
            at_or_before = bool(value <= th[pointer])
            close = bool(value.allclose(th[pointer]))
            if at_or_before or close:
                break
            pointer += 1
        rows.append(pointer + len(fresh))
        if not close:
            fresh.append(value.unsqueeze(0))
    batch = x.shape[:-2]
    t_dev = t.to(x.device)
    if fresh:                                                  # merge the new times in as missing observations
        merged, order = torch.cat([th, *fresh]).sort()
        missing = torch.full((*batch, 1, x.size(-1)), float("nan"), dtype=x.dtype, device=x.device)
        x = torch.cat([x, missing], dim=-2)[..., order.clamp(0, x.size(-2)).to(x.device), :]
        t_dev = merged.to(x.device)
    x = linear_interpolation_coeffs(x, t_dev)                  # fills the NaNs (the new rows and any in the data)
    L, C = x.size(-2), x.size(-1)
    src = x.detach().contiguous()
    B = src.numel() // (L * C)
    n_windows = len(rows) - 1
    rows_dev = torch.tensor(rows, dtype=torch.int64).to(x.device)
    scale = (new_t[1:] - new_t[:-1]) if version == 0 else torch.ones(n_windows, dtype=th.dtype)
    scale_dev = scale.to(device=x.device, dtype=x.dtype).contiguous()
    table = []
    for word in words:
        flat = 0
        for letter in word:
            flat = flat * C + letter
        table.append((len(word), flat))
    words_dev = torch.tensor(table, dtype=torch.int32).to(x.device)
    out = torch.empty(*batch, n_windows + 1, len(words), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    _lib.check(lib.cde_logsig_windows(_lib.ptr(src), _lib.ptr(rows_dev), _lib.ptr(scale_dev), _lib.ptr(words_dev),
                                      _lib.ptr(out), B, L, C, depth, n_windows, len(words), _lib.dtype_enum(x.dtype),
                                      _lib.stream_ptr(x.device)), "cde_logsig_windows")
    if version == 0:
        return out, new_t.to(x.device)
    return out


def logsig_windows(x, depth, window_length, t=None):
    """Logsignatures over windows of length ``window_length``, accumulated (reference log_ode.py:107-133): the values of
    the transformed path at times 0, 1, 2, ...; feed them to ``linear_interpolation_coeffs`` / ``LinearInterpolation``."""
    return _windows(x, depth, window_length, t, 1)


def logsignature_windows(x, depth, window_length, t=None):
    """Deprecated variant of the reference (log_ode.py:78-104): every window's logsignature is scaled by the window
    length and the window times are returned as well."""
    return _windows(x, depth, window_length, t, 0)
