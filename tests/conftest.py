import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """One device context for the whole GPU session (fails loudly without a gfx950 GPU)."""
    from mad_amd import _lib
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    L = _lib.Lib(0)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    L.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    L.set_eqsp(1, e16.sphere_eqsp)
    yield L
    L.close()
