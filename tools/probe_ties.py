#!/usr/bin/env python3
"""How many samples of a base-octave row lie next to a nearest-voxel tie (Descriptor.py:132-149), and for which rotations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402


def main():
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    lib = _lib.Lib(0)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS["c3"])
    src = subs[0]
    s = lib.set_build(src.slots, src.coords, src.octave, src.subv, src.index)
    d = s.download()
    rows = np.nonzero(src.octave[d["anchor"]] == 1)[0]
    l = np.arange(16) - 7.5
    L = np.stack(np.meshgrid(l, l, l, indexing="ij"), -1).reshape(-1, 3)
    near = []
    for r in rows:
        inv = np.linalg.inv(d["R"][r])
        a = (L @ inv.T).astype(np.float32)
        fr = a - np.floor(a)
        near.append(int((np.abs(fr - 0.5).min(axis=1) <= 2e-4).sum()))
    near = np.array(near)
    print("base-octave rows %d; samples next to a tie per row: p50 %d p90 %d p99 %d max %d; rows with > 1000: %d" % (
        len(rows), np.median(near), np.percentile(near, 90), np.percentile(near, 99), near.max(), (near > 1000).sum()))
    for r in rows[np.argsort(-near)[:6]]:
        print("row %d main %d sec %d near %d\n%s" % (r, d["main"][r], d["sec"][r], near[list(rows).index(r)], np.array2string(np.linalg.inv(d["R"][r]), precision=17)))


if __name__ == "__main__":
    main()
