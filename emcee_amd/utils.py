"""Small helpers (reference ``utils.py``: only what the hot path's driver needs)."""
import warnings

__all__ = ["deprecation_warning"]


def deprecation_warning(msg):
    warnings.warn(msg, category=DeprecationWarning, stacklevel=2)
