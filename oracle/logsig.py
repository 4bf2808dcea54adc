"""Oracle: logsignatures and the log-ODE windowing (CPU, torch).  TEST INFRASTRUCTURE.

The reference's ``torchcde/log_ode.py`` delegates the arithmetic to the third-party package ``signatory``
(``signatory.Logsignature(depth)``, ``signatory.logsignature_channels``; call sites log_ode.py:53,57,59), which is
neither vendored nor installed here.  **PARITY UNPINNED** against the real package: this file restates the published
definition -- the logsignature of a piecewise-linear path is the tensor-algebra logarithm of its signature (Chen's
identity over the linear pieces), and signatory's default ``mode="words"`` reports the coefficients of the Lyndon
words (ordered by length, then lexicographically) in that expanded logarithm -- and is anchored by mathematics in
tests/test_oracle.py: increments (depth 1), Levy areas (depth 2), the Baker-Campbell-Hausdorff expansion of two
straight segments expanded bracket by bracket (depth 3), invariance under re-parametrisation, Chen consistency.

The windowing / cumulative-sum logic of ``logsig_windows`` IS pinned: ``oracle/make_golden.py`` runs the reference's
own ``torchcde.log_ode`` with this module registered as ``signatory`` (the same device used for ``torchdiffeq``).
"""
import torch


def lyndon_words(channels, depth):
    """Lyndon words over {0..channels-1} of length <= depth, ordered by length then lexicographically (Duval)."""
    words = []
    w = [-1]
    while w:
        w[-1] += 1
        words.append(tuple(w))
        m = len(w)
        while len(w) < depth:
            w.append(w[len(w) - m])
        while w and w[-1] == channels - 1:
            w.pop()
    words = [u for u in words if len(u) <= depth]
    return sorted(words, key=lambda u: (len(u), u))


def logsignature_channels(channels, depth):
    """signatory.logsignature_channels: the number of Lyndon words (Witt's formula)."""
    return len(lyndon_words(channels, depth))


def _outer(a, b):
    return (a.unsqueeze(-1) * b.reshape(*b.shape[:-1], 1, b.shape[-1])).reshape(*a.shape[:-1], -1)


def signature_levels(path, depth):
    """Signature levels 1..depth of a piecewise-linear path (..., length, channels), each level flattened to
    (..., channels**k).  Chen: S <- S (x) exp(d) for every increment d."""
    lead, c = path.shape[:-2], path.size(-1)
    levels = [torch.zeros(*lead, c ** k, dtype=path.dtype) for k in range(1, depth + 1)]
    for i in range(path.size(-2) - 1):
        d = path[..., i + 1, :] - path[..., i, :]
        exp = [d]
        for k in range(2, depth + 1):
            exp.append(_outer(exp[-1], d) / k)
        new = []
        for k in range(1, depth + 1):                       # level k of S (x) exp(d)
            acc = levels[k - 1] + exp[k - 1]
            for j in range(1, k):
                acc = acc + _outer(levels[j - 1], exp[k - j - 1])
            new.append(acc)
        levels = new
    return levels


def log_levels(levels):
    """Tensor-algebra logarithm of 1 + S, level by level:  log(1+S) = sum_n (-1)^(n+1) S^n / n."""
    depth = len(levels)
    out = [lv.clone() for lv in levels]
    power = [lv.clone() for lv in levels]                   # S^n, levels 1..depth (level k of S^n is 0 for k < n)
    for n in range(2, depth + 1):
        nxt = [torch.zeros_like(lv) for lv in levels]
        for k in range(n, depth + 1):                       # (S^(n-1) (x) S)_k = sum_j S^(n-1)_j (x) S_(k-j)
            for j in range(n - 1, k):
                nxt[k - 1] = nxt[k - 1] + _outer(power[j - 1], levels[k - j - 1])
        power = nxt
        for k in range(n, depth + 1):
            out[k - 1] = out[k - 1] + ((-1) ** (n + 1)) * power[k - 1] / n
    return out


def logsignature(path, depth):
    """signatory.logsignature(path, depth) in its default "words" mode: (..., logsignature_channels)."""
    c = path.size(-1)
    logs = log_levels(signature_levels(path, depth))
    picks = []
    for word in lyndon_words(c, depth):
        flat = 0
        for letter in word:
            flat = flat * c + letter
        picks.append(logs[len(word) - 1][..., flat])
    return torch.stack(picks, dim=-1)


class Logsignature:
    """signatory.Logsignature(depth): callable on (batch, length, channels)."""

    def __init__(self, depth):
        self.depth = depth

    def __call__(self, path):
        return logsignature(path, self.depth)


def as_signatory_module():
    """A stand-in for the ``signatory`` package exposing exactly what torchcde/log_ode.py uses."""
    import types
    module = types.ModuleType("signatory")
    module.Logsignature = Logsignature
    module.logsignature_channels = logsignature_channels
    module.logsignature = logsignature
    return module


def window_plan(t, window_length):
    """log_ode.py:18-40: the window end times ``new_t`` (float tensor), for each of them its row index in the series
    AFTER the new times have been merged in, and the new times that are not already observation times."""
    timespan = t[-1] - t[0]
    pieces = int((timespan / window_length).ceil().item())
    new_t = torch.linspace(t[0].item(), (t[0] + pieces * window_length).item(), pieces + 1, dtype=t.dtype)
    new_t = torch.min(new_t, t.max())
    pointer, fresh, rows = 0, [], []
    for value in new_t:
        while True:
            at_or_before = bool(value <= t[pointer])
            close = bool(value.allclose(t[pointer]))
            if at_or_before or close:
                break
            pointer += 1
        rows.append(pointer + len(fresh))
        if not close:
            fresh.append(value.unsqueeze(0))
    return new_t, rows, fresh


def logsig_windows(x, depth, window_length, t=None, version=1):
    """log_ode.py:15-75 (``logsig_windows`` = version 1, ``logsignature_windows`` = version 0, which also returns the
    window times): merge the window boundaries into the series as missing observations, fill linearly, take the
    logsignature of every window, prepend the first observation and accumulate."""
    from . import interp
    t = interp.check_path(x, t)
    new_t, rows, fresh = window_plan(t, window_length)
    batch = x.shape[:-2]
    if fresh:
        merged, order = torch.cat([t, *fresh]).sort()
        missing = torch.full((*batch, 1, x.size(-1)), float("nan"), dtype=x.dtype)
        x = torch.cat([x, missing], dim=-2)[..., order.clamp(0, x.size(-2)), :]
        t = merged
    x = interp.linear_coeffs(x, t)
    flat = x.reshape(-1, x.size(-2), x.size(-1))
    first = torch.zeros(*batch, logsignature_channels(x.size(-1), depth), dtype=x.dtype)
    first[..., :x.size(-1)] = x[..., 0, :]
    pieces = [first]
    for lo, hi, t_lo, t_hi in zip(rows[:-1], rows[1:], new_t[:-1], new_t[1:]):
        piece = logsignature(flat[..., lo:hi + 1, :], depth).view(*batch, -1)
        if version == 0:
            piece = piece * (t_hi - t_lo)
        pieces.append(piece)
    out = torch.stack(pieces, dim=-2).cumsum(dim=-2)
    return (out, new_t) if version == 0 else out
