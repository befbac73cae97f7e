"""Model registry resolved by name, like ``getattr(fuxictr.pytorch.models, params["model"])`` (run_expid.py:75)."""
from .model import RAT_m2  # noqa: F401
from .variants import RAT_m0, RAT_m1, RAT_m3  # noqa: F401

__all__ = ["RAT_m0", "RAT_m1", "RAT_m2", "RAT_m3"]
