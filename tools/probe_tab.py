#!/usr/bin/env python3
"""Descriptors of a workload's map with the table classifier of the 4-byte texels (default) against MAD_NO_TAB=1 (the former
float32 tier on the 16-byte texels): tools/probe_tab.py <workload> <out.npz>; run twice, then compare with --compare a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    da, db = a["dsc"].astype(int), b["dsc"].astype(int)
    bad = np.flatnonzero(np.any(da != db, axis=1))
    print("rows", len(da), "differing rows", len(bad))
    for r in bad[:12]:
        w = np.flatnonzero(da[r] != db[r])
        print(" row", r, "anchor", a["anchor"][r], "entries", [(int(i) // 16, int(i) % 16, int(da[r, i]), int(db[r, i])) for i in w[:8]], "sum", da[r].sum(), db[r].sum())
    sys.exit(0)

import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402
from mad_amd.orient_tables import orientation_matrices      # noqa: E402

lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS[sys.argv[1]], 0)
s = lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index)
d = s.download()
np.savez(sys.argv[2], dsc=d["dsc"], anchor=d["anchor"])
print("rows", len(d["dsc"]))
