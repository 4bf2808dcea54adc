"""torchcde_amd -- the MI355X-native Neural-CDE hot path behind torchcde's API.

Drop-in names (reference ``torchcde/__init__.py:1-7``): ``hermite_cubic_coefficients_with_backward_differences``,
``natural_cubic_coeffs`` / ``natural_cubic_spline_coeffs``, ``linear_interpolation_coeffs``, ``CubicSpline`` (+ ``NaturalCubicSpline`` alias), ``LinearInterpolation``,
``InterpolationBase``, ``TupleControl``, ``logsig_windows`` / ``logsignature_windows``, ``cdeint``.  Everything numerical runs in hand-written HIP kernels (gfx950) loaded from
``libcde_mi355x.so`` through the C ABI of ``include/cde_mi355x.h``; there is no eager or CPU fallback.
"""
from ._lib import build, load, SO_PATH, set_option, get_option, tuning
from .paths import (InterpolationBase, CubicSpline, NaturalCubicSpline, LinearInterpolation,
                    hermite_cubic_coefficients_with_backward_differences, linear_interpolation_coeffs,
                    natural_cubic_coeffs, natural_cubic_spline_coeffs)
from .fields import LinearCDEFunc
from .cdeint import cdeint
from . import misc  # noqa: F401  (torchcde.misc.forward_fill)
from .misc import TupleControl
from .log_ode import logsig_windows, logsignature_windows

__version__ = "0.1.0"
