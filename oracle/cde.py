"""Oracle: the CDE -> ODE reduction and solver front-end (CPU).  TEST INFRASTRUCTURE.

Restates reference ``torchcde/solver.py``:
  ControlledField  <- _VectorField.forward            solver.py:117-135  (f(t,z) @ dX/dt, or func.prod)
  cdeint           <- cdeint                           solver.py:144-245  (tolerance defaults :195-203,
                                                        compat probe :205, dispatch :224-232, permute :234-243)
Only the torchdiffeq backend and tensor state are restated (tuple state and the torchsde
backend are outside SURVEY section 8).  Integration itself is ``oracle.odeint`` (PARITY UNPINNED,
see that file); this wrapper is pinned by running the real ``torchcde.cdeint`` over the same
integrator in ``oracle/make_golden.py``.
"""
import torch

from . import odeint as _ode


class ControlledField(torch.nn.Module):
    def __init__(self, X, func):
        super().__init__()
        self.X = X
        self.func = func
        self.uses_prod = hasattr(func, "prod")

    def forward(self, t, z):
        dX = self.X.derivative(t)                                   # (..., C)
        if self.uses_prod:
            return self.func.prod(t, z, dX)
        F = self.func(t, z)                                         # (..., H, C)
        return (F @ dX.unsqueeze(-1)).squeeze(-1)                   # (..., H)


def cdeint(X, func, z0, t, adjoint=True, integrators=None, **kwargs):
    """``integrators``: (odeint, odeint_adjoint) to use instead of oracle.odeint's -- oracle/pin_torchdiffeq.py passes the
    REAL torchdiffeq functions here when that package is importable."""
    kwargs.setdefault("atol", 1e-6)
    kwargs.setdefault("rtol", 1e-4)
    if adjoint:
        kwargs.setdefault("adjoint_atol", kwargs["atol"])
        kwargs.setdefault("adjoint_rtol", kwargs["rtol"])
    if not hasattr(X, "derivative"):
        raise ValueError("X must have a 'derivative' method.")
    if not isinstance(z0, torch.Tensor):
        raise NotImplementedError("oracle: tuple state is outside the hot-path scope")
    field = ControlledField(X, func)
    plain, with_adjoint = (_ode.odeint, _ode.odeint_adjoint) if integrators is None else integrators
    solve = with_adjoint if adjoint else plain
    out = solve(field, z0, t, **kwargs)                             # (T, ..., H)
    lead = range(1, out.dim() - 1)
    return out.permute(*lead, 0, -1)                                # (..., T, H)
