from mad_amd.eqsp.eqsp import *  # noqa: F401,F403
from mad_amd.eqsp.eqsp import EQSP_Sphere  # noqa: F401
