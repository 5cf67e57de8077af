"""Small rigid-body helpers kept on the host (float64 numpy).

Same names and conventions as the reference's mad/math_utils.py:5-53; the
scoring functions there (bc/mcc/precision/f1, :58-142) are dead code and are
not carried over.
"""
import numpy as np


def unit_vector(vec):
    """vec / |vec|; a vector that cannot be normalised is returned unchanged (math_utils.py:5-13)."""
    v = np.asarray(vec, dtype=np.float64)
    n = np.sqrt(np.dot(v, v))
    if not np.isfinite(n) or n == 0:
        print("MaD> ERROR: can't normalize vec", vec)
        return vec
    return v / n


def euler_rod_mat(axis, angle):
    """Euler-Rodrigues matrix, element for element as math_utils.py:15-27."""
    axis = np.asarray(axis, dtype=np.float64)
    a = np.cos(angle / 2.0)
    b, c, d = -axis * np.sin(angle / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])


def get_rototrans_SVD(mobile, reference):
    """Kabsch superposition in the row-vector convention x' = x @ R + T (math_utils.py:29-53)."""
    mobile = np.asarray(mobile, dtype=np.float64)
    reference = np.asarray(reference, dtype=np.float64)
    if mobile.shape != reference.shape or mobile.ndim != 2 or mobile.shape[1] != 3:
        raise Exception("Descript> ERROR: Coordinates mismatch for SVD")
    n = reference.shape[0]
    cm = sum(mobile) / n
    cr = sum(reference) / n
    u, _, vt = np.linalg.svd(np.dot((mobile - cm).T, reference - cr))
    R = np.dot(vt.T, u.T).T
    if np.linalg.det(R) < 0:      # reflection: flip the weakest axis
        vt[2] = -vt[2]
        R = np.dot(vt.T, u.T).T
    return R, cr - np.dot(cm, R)


def polar_to_cart(theta, phi):
    return np.array([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)])
