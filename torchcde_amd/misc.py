"""``torchcde.misc`` names that sit on the native path (reference torchcde/misc.py:103-126)."""
from .paths import forward_fill  # noqa: F401
