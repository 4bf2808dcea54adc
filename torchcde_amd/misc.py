"""``torchcde.misc`` names that sit on the native path."""
import torch

from .paths import forward_fill  # noqa: F401  (reference torchcde/misc.py:103-126)


class TupleControl(torch.nn.Module):
    """Several controls over one interval presented as a single control whose ``evaluate`` / ``derivative`` return
    tuples -- the control type for tuple-valued state (reference torchcde/misc.py:129-164, README "stacking")."""

    def __init__(self, *controls):
        super().__init__()
        if not controls:
            raise ValueError("Expected one or more controls to batch together.")
        first = controls[0]
        for other in controls[1:]:
            if not torch.equal(other.interval, first.interval):
                raise ValueError("Can only batch togehter controls over the same interval.")
        shared = all(other.grid_points.shape == first.grid_points.shape and torch.equal(other.grid_points, first.grid_points)
                     for other in controls[1:])
        self._shared_grid = first.grid_points if shared else None
        self._span = first.interval
        self.controls = torch.nn.ModuleList(controls)

    @property
    def interval(self):
        return self._span

    @property
    def grid_points(self):
        if self._shared_grid is None:
            raise RuntimeError("Batch of controls have different grid points.")
        return self._shared_grid

    def evaluate(self, t):
        return tuple(c.evaluate(t) for c in self.controls)

    def derivative(self, t):
        return tuple(c.derivative(t) for c in self.controls)

    def _second_derivative(self, t):
        return tuple(c._second_derivative(t) for c in self.controls)
