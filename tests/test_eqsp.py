"""EQSP tables: generator == shipped tables == the reference's bytes; host class contract."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

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


@pytest.mark.parametrize("Z", [16, 112])
def test_zone_bounds_overlap_only_at_the_seam(Z):
    """What the guard band of the device's float32 classifier (MAD_EQSP_GUARD, mad_amd/csrc/mad_common.h) is derived from: inside a belt
    the zones' tabulated theta bounds meet exactly, except at the seam -- the last zone ends at its 4-decimal theta_max (e.g. 6.2832)
    while the first one, matched through theta + 2 pi (Orientator.py:313, 328-331), begins 1.47e-5 rad below it -- and the belts share
    their phi bounds.  A table with a wider overlap would need a wider guard."""
    import math
    t = np.asarray(EQSP_Sphere(Z).sphere_eqsp)      # theta_min, phi_min, theta_max, phi_max
    belts = {}
    for r in t:
        belts.setdefault((r[1], r[3]), []).append((r[0], r[2]))
    seams = []
    for zs in belts.values():
        zs.sort()
        for i, (lo, hi) in enumerate(zs):
            if len(zs) == 1:
                continue
            nxt = zs[(i + 1) % len(zs)][0] + (2 * math.pi if i == len(zs) - 1 else 0.0)
            if i == len(zs) - 1:
                seams.append(hi - nxt)
            else:
                assert hi == nxt
    assert len(seams) == len([z for z in belts.values() if len(z) > 1])
    assert all(0 < s < 1.5e-5 for s in seams), seams
    phis = sorted(belts)
    assert all(a[1] == b[0] for a, b in zip(phis[:-1], phis[1:]))
