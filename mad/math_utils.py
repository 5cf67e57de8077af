"""Drop-in alias: the reference exposes this module as `mad.math_utils`; the implementation lives in `mad_amd.math_utils`."""
from mad_amd.math_utils import *  # noqa: F401,F403
from mad_amd import math_utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
