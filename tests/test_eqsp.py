"""EQSP tables: generator == shipped tables == the reference's bytes; host class contract."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

from mad_amd.eqsp import EQSP_Sphere

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_tables_match_generator_and_reference():
    with np.load(os.path.join(G, "g1_eqsp_math.npz")) as z:
        shas = {k: str(z[k]) for k in z.files if k.startswith("sha_")}
    tmp = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_eqsp_tables.py"), tmp])
    for stem in ("sphere", "centers"):
        for n in (16, 112):
            name = "%s_%d.txt" % (stem, n)
            shipped = open(os.path.join(ROOT, "mad_amd", "eqsp", name), "rb").read()
            assert shipped == open(os.path.join(tmp, name), "rb").read()
            assert hashlib.sha256(shipped).hexdigest() == shas["sha_%s_%d" % (stem, n)]


def test_host_class_contract():
    with np.load(os.path.join(G, "g1_eqsp_math.npz")) as z:
        for n in (16, 112):
            e = EQSP_Sphere(n)
            np.testing.assert_array_equal(e.sphere_eqsp, z["bounds_%d" % n])
            np.testing.assert_array_equal(e.p_centers_eqsp, z["centers_%d" % n])
            np.testing.assert_array_equal(e.c_centers_eqsp, z["c_centers_%d" % n])
            np.testing.assert_array_equal([len(b) for b in e.belt_l], z["belt_sizes_%d" % n])
            assert e.belt_of_idx(0) == 0 and e.belt_of_idx(n - 1) == len(e.belt_l) - 1
            assert e.area(3).shape == (4,) and e.p_center(3).shape == (2,) and e.c_center(3).shape == (3,)
