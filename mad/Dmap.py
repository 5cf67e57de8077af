"""Drop-in alias: the reference exposes this module as `mad.Dmap`; the implementation lives in `mad_amd.Dmap`."""
from mad_amd.Dmap import *  # noqa: F401,F403
from mad_amd import Dmap as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
