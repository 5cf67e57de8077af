#!/usr/bin/env python3
"""Host model of the table classifier of the 4-byte texels (mad_common.h: EqspTabLds, eqsp_tab32, mad_tex4_encode; tables as
mad_set_eqsp builds them): random unit directions, quantised to 3 x 10 bits, rotated, classified -- every decided sample must
carry the zone of the exact float64 classification of the unquantised direction.  Prints the undecided fraction.
    python tools/check_tab_classifier.py [n_directions]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_amd import synth      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402

NZ, NP, G = 2048, 2048, 4e-3      # MAD_TAB_ZBINS, MAD_TAB_PBINS, MAD_TAB_GUARD


def tables(B):
    Z = len(B)
    th_lo, ph_lo_z, th_hi, ph_hi_z = B[:, 0], B[:, 1], B[:, 2], B[:, 3]
    belts, prev = [], None
    for a in range(Z):
        if prev is None or ph_lo_z[a] != prev:
            belts.append(dict(first=a, count=0, ph_lo=ph_lo_z[a], ph_hi=ph_hi_z[a]))
            prev = ph_lo_z[a]
        belts[-1]["count"] += 1
    zbelt = np.full(NZ, 255, np.uint8)
    for k in range(NZ):
        zlo, zhi = -1 + (k - 1) * (2 / NZ), -1 + (k + 2) * (2 / NZ)
        if zlo <= -1 or zhi >= 1:
            continue
        pmin, pmax = np.arccos(zhi) - G, np.arccos(zlo) + G
        for bi, b in enumerate(belts):
            if pmin > b["ph_lo"] and pmax < b["ph_hi"]:
                zbelt[k] = bi

    def theta_of(p):
        if p <= 2:
            xr = 1 - p
            return np.arctan2(1 - abs(xr), xr)
        xr = p - 3
        return np.arctan2(-(1 - abs(xr)), xr) + 2 * np.pi

    ptab = np.full((4, NP), 255, np.uint8)
    for bi, b in enumerate(belts):
        if b["count"] == 1:
            ptab[bi, :] = b["first"]
            continue
        gt = G / min(np.sin(b["ph_lo"]), np.sin(b["ph_hi"]))
        for k in range(2, NP - 2):
            t0, t1 = theta_of((k - 1) * 4 / NP) - gt, theta_of((k + 2) * 4 / NP) + gt
            for a in range(b["first"], b["first"] + b["count"]):
                if (t0 > th_lo[a] and t1 < th_hi[a]) or (t0 + 2 * np.pi > th_lo[a] and t1 + 2 * np.pi < th_hi[a]):
                    ptab[bi, k] = a
    return zbelt, ptab


def exact(d, B):      # Descriptor.py:158-187: default zone 0, the last matching zone wins
    th = np.arctan2(d[:, 1], d[:, 0])
    th = np.where(th < 0, th + 2 * np.pi, th)
    sth = th + 2 * np.pi
    ph = np.arccos(np.clip(d[:, 2], -1, 1))
    zone = np.zeros(len(d), int)
    for a in range(len(B)):
        m = (((th > B[a, 0]) & (th < B[a, 2])) | ((sth > B[a, 0]) & (sth < B[a, 2]))) & (ph > B[a, 1]) & (ph < B[a, 3])
        zone[m] = a
    return zone


def classify(q, R, zbelt, ptab):
    f = R.astype(np.float32).copy()
    f[2] *= np.float32(1 / 511)
    r = q @ f.T
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    b = zbelt[np.clip(np.floor((z + 1) * (NZ / 2)).astype(int), 0, NZ - 1)]
    xr = x / np.maximum(np.abs(x) + np.abs(y), 1e-30)
    p = np.where(y >= 0, 1 - xr, 3 + xr)
    zn = ptab[b & 3, np.clip(np.floor(p * (NP / 4)).astype(int), 0, NP - 1)].astype(int)
    return np.where((b == 255) | (zn == 255), -1, zn)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    B = EQSP_Sphere(16).sphere_eqsp
    zbelt, ptab = tables(B)
    rng = np.random.default_rng(0)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = np.clip(np.rint(v * 511), -511, 511).astype(np.float32)
    bad = und = 0
    blk = 100000
    for s0 in range(0, n, blk):
        R = synth.random_rotation(rng) if s0 else np.array([[0.5, -0.8660254, 0], [0.8660254, 0.5, 0], [0, 0, 1.0]])
        sl = slice(s0, s0 + blk)
        ex = exact(v[sl].astype(np.float64) @ R.T, B)
        zn = classify(q[sl], R, zbelt, ptab)
        und += int(np.sum(zn < 0))
        bad += int(np.sum((zn >= 0) & (zn != ex)))
    print("directions %d  undecided %.4f  decided wrongly %d" % (n, und / n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
