"""The 3x3 matrices orientation assignment derives from the EQSP zone centres.

A main bin `a` fixes `to_dom_mat` (Orientator.py:198-205: rotate the zone centre onto
+z; identity for the pole zone, :211) and a secondary bin `s` fixes `adj_sec_mat`
(Orientator.py:253-263: rotate about z so that the zone's azimuth becomes that of the
first zone of its belt).  Both depend on the zone index only, so they are tabulated
once on the host with the reference's own float64 formulas and handed to the device
(mad_set_eqsp), which then forms Rfinal = adj_sec_mat @ to_dom_mat (Orientator.py:105).
"""
import numpy as np

from .math_utils import euler_rod_mat, unit_vector


def to_dom_mat(eqsp, main_bin):
    if main_bin == 0:
        return np.identity(3)
    c = unit_vector(eqsp.c_center(main_bin))
    angle = np.arccos(np.clip(np.dot(c, [0, 0, 1]), -1.0, 1.0))
    axis = unit_vector(np.cross(c, [0, 0, 1]))
    return np.array(euler_rod_mat(axis, angle))


def adj_sec_mat(eqsp, sec_bin):
    first = eqsp.belt_l[eqsp.belt_of_idx(sec_bin)][0]
    ftheta = -1 * (eqsp.p_center(sec_bin)[0] - eqsp.p_center(first)[0])
    return np.array(euler_rod_mat([0, 0, 1], ftheta))


def orientation_matrices(eqsp):
    """(to_dom[Z,3,3], adj_sec[Z,3,3]) for every zone of `eqsp`."""
    dom = np.stack([to_dom_mat(eqsp, a) for a in range(eqsp.size)])
    adj = np.stack([adj_sec_mat(eqsp, a) for a in range(eqsp.size)])
    return dom, adj
