"""Equal-area sphere partition tables (host side).

Mirror of the reference's `EQSP_Sphere` (mad/eqsp/eqsp.py:12-87): same attribute
names (`sphere_eqsp`, `p_centers_eqsp`, `c_centers_eqsp`, `belt_l`, `max_belt_l`,
`equator_idx`, `size`) and accessors, so code written against the reference keeps
working.  Differences: the tables are resolved relative to this package instead
of the process CWD (eqsp.py:16,26 open "mad/eqsp/..."), and the unused
`feature_dist_thresh` is computed with numpy instead of scikit-learn.

The tables themselves are produced by tools/gen_eqsp_tables.py (Leopardi's
construction) and are byte-identical to the reference's.
"""
import os
from math import cos, sin

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def _load(size):
    if size not in _CACHE:
        paths = [os.path.join(_DIR, "%s_%i.txt" % (stem, size)) for stem in ("sphere", "centers")]
        for p in paths:
            if not os.path.exists(p):
                raise FileNotFoundError("MaD> no EQSP table for size %i (%s); only 16 and 112 ship" % (size, p))
        _CACHE[size] = (np.loadtxt(paths[0], dtype=np.float64, ndmin=2), np.loadtxt(paths[1], dtype=np.float64, ndmin=2))
    return _CACHE[size]


class EQSP_Sphere(object):
    def __init__(self, size=112):
        bounds, centers = _load(size)
        self.size = size
        # rows: theta_min, phi_min, theta_max, phi_max
        self.sphere_eqsp = bounds.copy()
        # rows: theta, phi and the matching unit vectors
        self.p_centers_eqsp = centers.copy()
        self.c_centers_eqsp = np.array([[sin(p) * cos(t), sin(p) * sin(t), cos(p)] for t, p in centers])

        # belts: consecutive zones sharing phi_min (eqsp.py:37-46)
        belts = []
        last = None
        for idx in range(size):
            lo = self.sphere_eqsp[idx, 1]
            if last is None or lo != last:
                belts.append([])
                last = lo
            belts[-1].append(idx)
        self.belt_l = belts
        self.max_belt_l = 0
        seen = 0
        for b in belts:
            if len(b) > self.max_belt_l:
                self.max_belt_l = len(b)
                self.equator_idx = seen
            seen += len(b)
        self.belt_first = np.zeros(size, dtype=np.int32)
        for b in belts:
            self.belt_first[b] = b[0]

        # mean nearest-neighbour distance between centres * 0.1 (eqsp.py:62-64; unused downstream)
        d = np.linalg.norm(self.c_centers_eqsp[:, None, :] - self.c_centers_eqsp[None, :, :], axis=-1)
        np.fill_diagonal(d, np.inf)
        self.feature_dist_thresh = float(np.average(d.min(axis=1)) * 0.1)

    def p_center(self, idx):
        return self.p_centers_eqsp[idx]

    def c_center(self, idx):
        return self.c_centers_eqsp[idx]

    def area(self, idx):
        return self.sphere_eqsp[idx]

    def dist_thresh(self):
        return self.feature_dist_thresh

    def belt_indices(self, idx):
        return self.belt_l[idx]

    def belt_of_idx(self, idx):
        for i, b in enumerate(self.belt_l):
            if idx in b:
                return i
        return None

    def belt_equator_bounds(self):
        return self.sphere_eqsp[self.equator_idx]
