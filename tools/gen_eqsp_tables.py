#!/usr/bin/env python3
"""Generate the EQSP (equal-area sphere partition) zone tables used by the hot path.

The reference ships four 4-decimal text tables (`mad/eqsp/sphere_{16,112}.txt`,
`mad/eqsp/centers_{16,112}.txt`, loaded at `mad/eqsp/eqsp.py:16-33`) that were
written once from P. Leopardi's recursive zonal equal-area partition (ETNA 25,
2006).  This script restates that published construction for S^2 and writes the
same tables from first principles, so the repo carries a generator instead of a
copy.  `tests/test_eqsp.py` checks that the shipped tables equal this script's
output and (via a committed sha256) the reference's bytes.

Row format, as the reference reader expects:
  sphere_N.txt  : theta_min phi_min theta_max phi_max   (theta = azimuth, phi = colatitude)
  centers_N.txt : theta phi
Zones are listed pole cap first, then collar by collar, then the south cap.  A
collar's sectors are rotated by Leopardi's running `circle_offset`; a sector
that straddles theta = 2*pi keeps theta_max > 2*pi (matched by the callers
through theta + 2*pi).
"""
import math
import os
import sys


def _cap_area(c):
    return 4.0 * math.pi * math.sin(c / 2.0) ** 2


def _cap_colat(area):
    return 2.0 * math.asin(math.sqrt(area / math.pi) / 2.0)


def _circle_offset(n_top, n_bot):
    return (1.0 / n_bot - 1.0 / n_top) / 2.0 + math.gcd(n_top, n_bot) / (2.0 * n_top * n_bot)


def eq_caps(n):
    """Colatitudes of the zone boundaries and the number of regions per zone."""
    if n == 1:
        return [math.pi], [1]
    if n == 2:
        return [math.pi / 2, math.pi], [1, 1]
    area = 4.0 * math.pi / n
    c_polar = _cap_colat(area)
    ideal_angle = math.sqrt(area)
    n_collars = max(1, int(round((math.pi - 2.0 * c_polar) / ideal_angle)))
    fit = (math.pi - 2.0 * c_polar) / n_collars
    ideal = [(_cap_area(c_polar + (j + 1) * fit) - _cap_area(c_polar + j * fit)) / area
             for j in range(n_collars)]
    n_regions = [1]
    disc = 0.0
    for r in ideal:
        k = int(math.floor(r + disc + 0.5))
        n_regions.append(k)
        disc += r - k
    n_regions.append(1)
    caps = []
    tot = 0
    for k in n_regions:
        tot += k
        caps.append(_cap_colat(tot * area))
    caps[-1] = math.pi
    return caps, n_regions


def eq_tables(n):
    """Return (bounds[n][4], centers[n][2]) in the reference's row order."""
    caps, n_regions = eq_caps(n)
    two_pi = 2.0 * math.pi
    bounds = [[0.0, 0.0, two_pi, caps[0]]]
    centers = [[0.0, 0.0]]
    offset = 0.0
    for ci in range(1, len(n_regions) - 1):
        k = n_regions[ci]
        top, bot = caps[ci - 1], caps[ci]
        for s in range(k):
            t0 = math.fmod(s * two_pi / k + two_pi * offset, two_pi)
            t1 = math.fmod((s + 1) * two_pi / k + two_pi * offset, two_pi)
            if t1 < t0:
                t1 += two_pi
            bounds.append([t0, top, t1, bot])
            centers.append([math.fmod((t0 + t1) / 2.0, two_pi), (top + bot) / 2.0])
        offset += _circle_offset(k, n_regions[ci + 1])
        offset -= math.floor(offset)
    bounds.append([0.0, caps[-2], two_pi, math.pi])
    centers.append([0.0, math.pi])
    return bounds, centers


def _fmt(rows):
    out = []
    for row in rows:
        vals = []
        for v in row:
            s = "%.4f" % v
            # a centre that wraps to exactly 2*pi prints as 0
            if s == "6.2832" and len(row) == 2:
                s = "0.0000"
            vals.append(s)
        out.append(" ".join(vals))
    return "\n".join(out) + "\n"


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    for n in (16, 112):
        b, c = eq_tables(n)
        with open(os.path.join(outdir, "sphere_%d.txt" % n), "w") as f:
            f.write(_fmt(b))
        with open(os.path.join(outdir, "centers_%d.txt" % n), "w") as f:
            f.write(_fmt(c))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "mad_amd", "eqsp"))
