"""Vector-field recognition for the fused solvers.

``cdeint`` accepts an arbitrary callable ``func(t, z) -> (..., H, C)`` (reference solver.py:159-165).
The fused kernels implement two families:
        f(t, z) = act( Linear(H, H*C)(z) ) viewed as (..., H, C),      act in {identity, tanh}
i.e. the README field (reference README.md:42-49) and ``example/irregular_data.py:36-46``, and
        f(t, z) = act( Linear(W, H*C)( relu( Linear(H, W)(z) ) ) ) viewed as (..., H, C)
i.e. ``example/time_series_classification.py:20-51`` (forward solves only).  A module is recognised by
*probing*, not by tracing source: it must own exactly one (two) nn.Linear and no other parameter; during
the compatibility evaluation ``func(t[0], z0)`` that the reference performs anyway, forward hooks record
the Linears' inputs and outputs, and the module's result must be BITWISE equal to the family's formula
applied to those recordings -- view/reshape never change values, so any other arithmetic (scaling, skip
connections, time dependence, another activation) fails the comparison and the module is refused (and
solved step by step) rather than mis-solved.

Value-dependent look-alikes (relu6 == relu while every pre-activation is below 6, hardtanh / clamp == identity
inside [-1, 1], dropout in eval mode ...) are excluded STRUCTURALLY, not by luck of the probed values: the
verification runs under a dispatch-mode recorder and refuses the module if it executes any operator outside
{matrix products, bias add, relu, tanh, pure layout ops}; it is repeated on an input scaled far outside the
data range and at a second time value.  The verdict is cached per module and re-used only while the module's
*fingerprint* -- class and ``forward`` function of every submodule (monkey-patching on the class or the
instance), train/eval flags and every plain Python attribute (e.g. a ``use_tanh`` switch) -- is unchanged.
What cannot be seen from one evaluation is Python control flow that branches on the VALUES of ``z``; such a
module must not be handed to the fused path (pass ``variant="generic"``-free step-wise solving by wrapping it
so that it owns a non-Linear parameter, or simply avoid value-dependent branches: the reference's own
adjoint also assumes ``func`` is a fixed differentiable map).
"""
import weakref

import torch
import torch.utils._python_dispatch

from . import _lib


class AffineField:
    """(linear module, activation enum) extracted from a user module."""
    kind = "affine"

    def __init__(self, linear, act):
        self.linear = linear
        self.linears = (linear,)
        self.act = act
        self.shapes = {}          # verified input signature -> output shape of func

    @property
    def weight(self):
        return self.linear.weight

    @property
    def bias(self):
        return self.linear.bias


class MLPField:
    """(hidden Linear, output Linear, final activation enum): Linear -> relu -> Linear -> act."""
    kind = "mlp2"

    def __init__(self, hidden, output, act):
        self.hidden, self.output = hidden, output
        self.linears = (hidden, output)
        self.act = act
        self.shapes = {}

    @property
    def weight(self):            # the layer that produces the (H, C) matrix
        return self.output.weight

    @property
    def bias(self):
        return self.output.bias


class LinearCDEFunc(torch.nn.Module):
    """Ready-made member of the fused family: ``Linear(H, H*C)`` (+ optional tanh) viewed (..., H, C)."""

    def __init__(self, input_channels, hidden_channels, tanh=False, **kwargs):
        super().__init__()
        self.input_channels = input_channels
        self.hidden_channels = hidden_channels
        self.use_tanh = tanh
        self.linear = torch.nn.Linear(hidden_channels, hidden_channels * input_channels, **kwargs)

    def forward(self, t, z):
        out = self.linear(z)
        if self.use_tanh:
            out = out.tanh()
        return out.view(*z.shape[:-1], self.hidden_channels, self.input_channels)


_verified = weakref.WeakKeyDictionary()   # func -> AffineField, after the two-time probe passed once

# operators a member of the fused families may execute (aten names at dispatch level); anything else -> step-wise
_ALLOWED_OPS = {
    "addmm", "mm", "bmm", "matmul", "linear", "add", "relu", "tanh",                      # arithmetic of the families
    "t", "transpose", "permute", "view", "_unsafe_view", "reshape", "_reshape_alias", "expand", "unsqueeze", "squeeze",
    "alias", "detach", "clone", "contiguous", "as_strided", "unflatten", "flatten", "select", "slice",   # layout only
}


class _OpRecorder(torch.utils._python_dispatch.TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.foreign = []

    def __torch_dispatch__(self, op, types, args=(), kwargs=None):
        name = op.overloadpacket.__name__ if hasattr(op, "overloadpacket") else str(op)
        if name not in _ALLOWED_OPS:
            self.foreign.append(name)
        return op(*args, **(kwargs or {}))


def _fingerprint(func):
    """Everything that can change what ``forward`` computes without touching a parameter value."""
    if not isinstance(func, torch.nn.Module):
        return None
    items = []
    for name, m in func.named_modules():
        plain = tuple(sorted((k, v) for k, v in vars(m).items()
                             if not k.startswith("_") and isinstance(v, (bool, int, float, str, type(None)))))
        items.append((name, type(m), type(m).forward, vars(m).get("forward"), m.training, plain))
    return tuple(items)


def _linears(func):
    """The module's nn.Linear layers if they hold ALL of its parameters (1 or 2 layers, with bias), else None."""
    if isinstance(func, LinearCDEFunc):
        return (func.linear,)
    if not isinstance(func, torch.nn.Module):
        return None
    linears = tuple(m for m in func.modules() if isinstance(m, torch.nn.Linear))
    if len(linears) not in (1, 2) or any(m.bias is None for m in linears) or len({id(m) for m in linears}) != len(linears):
        return None
    own = {id(p) for m in linears for p in m.parameters()}
    if any(id(p) not in own for p in func.parameters()):
        return None       # other trainable tensors take part: not one of the fused families
    return linears


def _evaluate_recording(func, linears, t, z):
    """(result, recorded Linear calls, names of operators outside the families' whitelist)"""
    calls = []
    handles = [m.register_forward_hook(lambda mod, inp, out: calls.append((mod, inp[0], out))) for m in linears]
    recorder = _OpRecorder()
    try:
        with torch.no_grad(), recorder:
            system = func(t, z)
    finally:
        for handle in handles:
            handle.remove()
    return system, calls, recorder.foreign


def _final_activation(system, last_out):
    if system.numel() != last_out.numel():
        return None
    flat = system.reshape(-1)
    if torch.equal(flat, last_out.reshape(-1)):
        return _lib.ACT_NONE
    if torch.equal(flat, last_out.tanh().reshape(-1)):
        return _lib.ACT_TANH
    return None


def _classify(system, calls, z):
    """(kind, act, ordered layers) if ``system`` is exactly one of the fused formulas applied to the recorded layer
    inputs/outputs, else None (bitwise comparison: view/reshape never change values, so any other arithmetic in
    ``func`` is detected)."""
    if not isinstance(system, torch.Tensor) or len(calls) not in (1, 2):
        return None
    first, seen_input, first_out = calls[0]
    if seen_input.shape != z.shape or not torch.equal(seen_input, z):
        return None
    if len(calls) == 1:
        act = _final_activation(system, first_out)
        return None if act is None else ("affine", act, (first,))
    second, hidden_in, second_out = calls[1]
    if second is first or hidden_in.shape != first_out.shape or not torch.equal(hidden_in, first_out.relu()):
        return None
    act = _final_activation(system, second_out)
    return None if act is None else ("mlp2", act, (first, second))


def probe(func, t0, z0):
    """Evaluate ``func(t0, z0)`` once (the compatibility probe the reference performs anyway,
    solver.py:47-53) while watching the module's nn.Linear layers.

    Returns ``(field_or_None, system)``.  The first time a module (in its current fingerprint) is seen it is also
    evaluated at a second time value and on an input far outside the data range."""
    linears = _linears(func)
    if linears is None:
        with torch.no_grad():
            return None, func(t0, z0)
    signature = (tuple(z0.shape), z0.dtype, str(z0.device), _fingerprint(func))
    try:
        known = _verified.get(func)
    except TypeError:
        known = None
    if known is not None and set(map(id, known.linears)) == set(map(id, linears)) and signature in known.shapes:
        # Verified before in this very state (same classes, same forward functions, same flags and plain attributes) on
        # an input of this shape/dtype/device: the structural facts do not change with the VALUES of z or the weights,
        # so the compatibility evaluation -- launches plus synchronising comparisons -- is not repeated.
        return known, torch.empty(known.shapes[signature], dtype=z0.dtype, device="meta")
    system, calls, foreign = _evaluate_recording(func, linears, t0, z0)
    found = None if foreign else _classify(system, calls, z0)
    if found is None or len(found[2]) != len(linears):
        return None, system
    kind, act, ordered = found
    # second time value (no time dependence) and an input scaled far outside the data range (saturating look-alikes
    # of identity / relu / tanh differ there even if they agreed on the data)
    other_t = t0.detach() + 0.8125
    system2, calls2, foreign2 = _evaluate_recording(func, linears, other_t, z0)
    if foreign2 or _classify(system2, calls2, z0) != found or not torch.equal(system2, system):
        return None, system
    far = z0.detach() * 37.0 + 11.0
    system3, calls3, foreign3 = _evaluate_recording(func, linears, t0, far)
    if foreign3 or _classify(system3, calls3, far) != found:
        return None, system
    if known is None or known.kind != kind or known.act != act or known.linears != ordered:
        known = AffineField(ordered[0], act) if kind == "affine" else MLPField(ordered[0], ordered[1], act)
        try:
            _verified[func] = known
        except TypeError:
            pass
    known.shapes = {k: v for k, v in known.shapes.items() if k[3] == signature[3]}     # drop other fingerprints
    known.shapes[signature] = tuple(system.shape)
    return known, system
