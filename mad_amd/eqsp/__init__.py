from .eqsp import EQSP_Sphere  # noqa: F401
