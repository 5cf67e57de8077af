"""Drop-in alias: the reference exposes this module as `mad.Descriptor`; the implementation lives in `mad_amd.Descriptor`."""
from mad_amd.Descriptor import *  # noqa: F401,F403
from mad_amd import Descriptor as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
