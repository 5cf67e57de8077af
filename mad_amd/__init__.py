"""mad_amd -- MI355X-native implementation of the MaD anchor-matching hot path.

Host-side mirror of the reference's Python interface (same class and method names)
over hand-written HIP kernels for gfx950, reached through the C-ABI declared in
include/mad_amd.h.  There is no CPU fallback: see mad_amd/_lib.py.
"""
__version__ = "0.1.0"
