"""Which code path a ``cdeint`` call takes -- as a table, not an if-lattice.

``cdeint`` (reference ``torchcde/solver.py:144-245``) accepts any callable ``func``, any torchdiffeq ``method`` / ``options``
and several ways of asking for gradients; the fused HIP kernels cover the combinations that matter and
everything else is solved step by step on the GPU (``stepwise.py``: host-driven, 10-1000x slower).  Round 2 decided this in
220 lines of nested conditions with silent cliffs.  Here the decision is a pure function of a small description of the
request (no tensors), so it can be enumerated exhaustively on the CPU (tests/test_host.py), queried after the fact
(``torchcde_amd.cdeint.last_dispatch``) and explained: every step-wise verdict carries its reason, and a RECOGNISED
vector field that falls off the fused paths raises one warning per (module class, reason).
"""
import collections
import threading
import warnings

# what cdeint knows about a call once the compatibility probe has run
Request = collections.namedtuple("Request", [
    "prod",              # func.prod exists (solver.py:48-53): the module multiplies by dX itself
    "kind",              # None: unrecognised module; "affine": act(Linear(z)); "mlp2": act(Linear(relu(Linear(z))))
    "tiles_ok",          # the field's shape / dtype fit some fused kernel (cde_rk4_supported; _mlp_fusable)
    "mfma_shape",        # affine field on the 32 x 8 MFMA tiles (float32, H <= 32, C <= 8, variant != "generic")
    "method",            # "rk4" | "dopri5" | any other torchdiffeq method name
    "adjoint",           # cdeint's adjoint flag
    "wants_grad",        # something requires a gradient through the solve
    "wants_t",           # ... the output times do
    "wants_control",     # ... the control's coefficient / knot tensors do (through adjoint_params; adjoint=False: any of them
                         #     that requires a gradient -- autograd reaches it through X.derivative)
    "params",            # adjoint_params: "default" | "own" (the field's parameters, all of them for mlp2, plus control
                         #                 tensors) | "foreign" (anything else)
    "adjoint_method_ok", # adjoint_method absent or equal to the forward method
    "options_ok",        # forward options within the fused kernels' set (rk4: step_size; dopri5: jump_t, safety, ifactor, dfactor)
    "adjoint_options_ok",  # the same for adjoint_options (dopri5: norm absent or "seminorm")
    "t_ok",              # t is strictly increasing
    "variant_generic",   # variant="generic" (tests: the independent kernels / the step-wise reference path)
    "shared",            # torchcde_amd.distributed.shared_step_control is active
    "narrow_control",    # the control has at most 8 channels (recorded; since round 5 no row depends on it: the two-layer sweeps
                         # accumulate control gradients on both of their tile layouts)
    "backprop_ok",       # affine: the field sits on the 32 x 8 tiles (float32, identity or tanh); mlp2: it fits the two-layer
                         # tiles -- what the reverse-mode sweeps (adjoint=False) take
    "identity",          # affine field without an activation (the README's): with backprop_ok, what the midpoint / euler forms
                         # of K2 / K3p take
    "control_block",     # the control tensors named in adjoint_params are the packed coefficient tensor the path was built from
                         # (README.md:251-270), optionally with the knot times (test/test_tricks.py:21-49) -- blocks of
                         # torchdiffeq's adjoint error norm which the adaptive backward K4a carries since round 6
])

Choice = collections.namedtuple("Choice", ["path", "reason"])

FUSED_PATHS = (
    "rk4",                    # K2 / K3 (+ split, wide, generic variants behind CDE_VARIANT_AUTO), incl. time / control gradients
    "dopri5_forward",         # K4 (MFMA, wide or generic attempt kernel), nothing to differentiate
    "dopri5_adjoint",         # K4 + K4a: the reference's default training call
    "mlp_rk4_forward",        # K2m
    "mlp_rk4_adjoint",        # K2m + K3m + factor reduction (optionally with control gradients)
    "mlp_dopri5_forward",     # K4 with the two-layer field
    "mlp_dopri5_adjoint",     # K4 + K4am: the reference examples' training call with their own model
    "rk4_backprop",           # adjoint=False: K2 storing its stage states + K3d, reverse mode through the solver's steps
    "fixed_grid",             # method='midpoint' / 'euler': K2 / K3p with two stages / one stage per step
    "mlp_rk4_backprop",       # adjoint=False for the two-layer field: K2m storing its stage states + K3m's sweep in reverse mode
)
STEPWISE = "stepwise"


def _stepwise(reason):
    return Choice(STEPWISE, reason)


def select_path(q):
    """The capability table.  Rows are tried top to bottom; the first that applies decides."""
    if q.prod:
        return _stepwise("func.prod multiplies by the control derivative itself")
    if q.kind is None:
        return _stepwise("the vector field is not of a recognised form (arbitrary module)")
    if not q.tiles_ok:
        return _stepwise("the field's shape or dtype is beyond the fused kernels' tiles")
    if not q.t_ok:
        return _stepwise("the output times are not strictly increasing")
    if q.method in ("midpoint", "euler"):
        # torchdiffeq's other fixed-grid methods (reference test/test_cdeint.py:49-63): K2 / K3p with two stages / one per step
        grads_ok = not q.wants_grad or (q.adjoint and q.adjoint_method_ok and q.adjoint_options_ok and q.params != "foreign")
        if (q.kind == "affine" and q.identity and q.backprop_ok and q.options_ok and not q.wants_t and not q.wants_control
                and grads_ok):
            return Choice("fixed_grid", "")
        return _stepwise("method %r is fused for the identity-activation affine field on the 32 x 8 tiles only (float32, "
                         "adjoint=True, no time / control gradients)" % (q.method,))
    if q.method not in ("rk4", "dopri5"):
        return _stepwise("method %r has no fused kernel (rk4, midpoint, euler and dopri5 have)" % (q.method,))
    if q.wants_grad and not q.adjoint:
        if q.backprop_ok and q.method == "rk4" and q.options_ok and not q.wants_t:
            return Choice("rk4_backprop" if q.kind == "affine" else "mlp_rk4_backprop", "")
        return _stepwise("adjoint=False with gradients: backpropagation through the solver's own operations (fused under rk4 "
                         "for the one-layer fields on the 32 x 8 tiles and for the two-layer field)")
    if not q.options_ok:
        return _stepwise("solver options outside the fused kernels' set")
    if q.wants_grad and not q.adjoint_method_ok:
        return _stepwise("adjoint_method differs from the forward method")
    if q.wants_grad and not q.adjoint_options_ok:
        return _stepwise("adjoint_options outside the fused kernels' set")
    if q.wants_grad and q.params == "foreign":
        return _stepwise("adjoint_params holds tensors other than the field's parameters and the control's")
    if q.kind == "mlp2":
        if q.variant_generic:
            return _stepwise("variant='generic' was requested (the step-wise reference path)")
        if not q.wants_grad:
            return Choice("mlp_rk4_forward" if q.method == "rk4" else "mlp_dopri5_forward", "")
        if q.wants_control:
            if q.method == "dopri5":
                if q.control_block and not q.shared:
                    return Choice("mlp_dopri5_adjoint", "")
                return _stepwise("control gradients through the adaptive backward: fused for the coefficient tensor the path "
                                 "was built from (optionally with its knot times) as the extra entries of adjoint_params, no "
                                 "shared step controller")
            return Choice("mlp_rk4_adjoint", "")
        if q.method == "rk4":
            return Choice("mlp_rk4_adjoint", "")
        return Choice("mlp_dopri5_adjoint", "")
    # one-layer (affine) fields
    if (q.wants_t or q.wants_control) and not q.mfma_shape:
        return _stepwise("time / control gradients outside the 32 x 8 MFMA tiles")
    if q.method == "rk4":
        return Choice("rk4", "")
    if not q.wants_grad:
        return Choice("dopri5_forward", "")
    if q.wants_control and not (q.control_block and q.mfma_shape and not q.shared):
        return _stepwise("control gradients through the adaptive backward: fused for the coefficient tensor the path was built "
                         "from (optionally with its knot times) as the extra entries of adjoint_params, one-layer fields on the "
                         "32 x 8 tiles, no shared step controller")
    if not q.mfma_shape:
        return _stepwise("the adaptive backward exists for the 32 x 8 MFMA tiles only (float32, H <= 32, C <= 8)")
    return Choice("dopri5_adjoint", "")


# ------------------------------------------------------------------------------------------ record + warnings
_local = threading.local()
_warned = set()
_warned_lock = threading.Lock()


def record(choice, request):
    """Remember the verdict of the calling thread's most recent cdeint call (`last()`)."""
    _local.last = (choice, request)


def last():
    """(Choice, Request) of this thread's most recent cdeint call, or None."""
    return getattr(_local, "last", None)


def warn_once(func, choice, request):
    """A recognised field that leaves the fused paths: one warning per (module class, reason)."""
    if choice.path != STEPWISE or request.kind is None or request.variant_generic:
        return
    key = (type(func), choice.reason)
    with _warned_lock:
        if key in _warned:
            return
        _warned.add(key)
    warnings.warn("torchcde_amd: this cdeint call runs step-wise (host-driven stepping on the GPU, typically 10-1000x "
                  "slower than the fused kernels) although %s is a vector field the fused kernels know: %s."
                  % (type(func).__name__, choice.reason), stacklevel=3)
