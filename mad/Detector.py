"""Drop-in alias: the reference exposes this module as `mad.Detector`; the implementation lives in `mad_amd.Detector`."""
from mad_amd.Detector import *  # noqa: F401,F403
from mad_amd import Detector as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
