#!/usr/bin/env python3
"""Host time of mad_set_build (enqueue only) with repeated and with fresh anchor lists: C3's map and one subunit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402
from mad_amd.orient_tables import orientation_matrices      # noqa: E402

lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS["c3"])
for name, st in (("map", the_map), ("subunit", subs[0])):
    sets = [_lib.DeviceSet(lib) for _ in range(4)]
    for fresh in (0, 1):
        for it in range(3):
            for i, s in enumerate(sets):
                lib.set_build(*st.anchor_list((it & 1) if fresh else 0), into=s)
        lib.synchronize()
        t = 0.0
        n = 0
        for it in range(40):
            for i, s in enumerate(sets):
                t0 = time.perf_counter()
                lib.set_build(*st.anchor_list((it & 1) if fresh else 0), into=s)
                t += time.perf_counter() - t0
                n += 1
            lib.synchronize()
        print("%-8s %4d anchors  %s lists: %.1f us per mad_set_build (host)" % (name, len(st.coords), "fresh" if fresh else "repeated", 1e6 * t / n))
