#!/usr/bin/env python3
"""Host emulation of the table tier on ONE row of a workload's map: which samples the table decides, and whether they agree with
the exact zones.  tools/probe_tab_row.py <workload> <row>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
exec(open("/tmp/tabcheck_tables.py").read()) if os.path.exists("/tmp/tabcheck_tables.py") else None
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402
from mad_amd.orient_tables import orientation_matrices      # noqa: E402
from oracle import oracle as O      # noqa: E402

lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS[sys.argv[1]], 0)
s = lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index)
d = s.download()
row = int(sys.argv[2])
a = d["anchor"][row]
octave = int(the_map.octave[a])
c = the_map.coords[a]
R = d["R"][row]
print("row", row, "anchor", a, "octave", octave, "coords", c, "R", R.tolist())
g = np.ascontiguousarray(np.moveaxis(the_map.ms.grad_list[octave], -1, 0), dtype=np.float32)
ref = O.describe(g[0], g[1], g[2], octave, c[None], R[None], e16.sphere_eqsp)[0]
got = d["dsc"][row]
print("device == oracle:", np.array_equal(ref, got), "diff entries", np.flatnonzero(ref != got)[:10])
one = lib.describe(the_map.slots[octave], octave, c[None], R[None])[0]
print("stage API == oracle:", np.array_equal(one, ref))
np.savez("gpurun_out/tab_row.npz", g=g[:, max(c[0]-40,0):c[0]+40, max(c[1]-40,0):c[1]+40, max(c[2]-40,0):c[2]+40], c=c, R=R, octave=octave, ref=ref, got=got,
         lo=np.array([max(c[0]-40,0), max(c[1]-40,0), max(c[2]-40,0)]))
if os.environ.get("MAD_LIB_PATH"):
    import ctypes as C
    dbg = np.zeros(4096 * 4, np.int32)
    one = lib.describe(the_map.slots[octave], octave, c[None], R[None])[0]      # row 0 of this launch is the probed one
    assert lib.dll.mad_debug_tab(dbg.ctypes.data_as(C.c_void_p)) == 0
    dbg = dbg.reshape(4096, 4)
    np.save("gpurun_out/tab_dbg.npy", dbg)
    print("zones", np.bincount(dbg[:, 0] + 2, minlength=19))
