#!/usr/bin/env python3
"""Soak: many overlapped steps of the benchmark's hot path, every step's top-k rows compared with the first step's.
    python tools/soak.py [workload] [steps] [in_flight]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    lib = _lib.Lib(0)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS[wl])
    groups = [[_lib.DeviceSet(lib) for _ in range(1 + len(subs))] for _ in range(depth)]
    lib.set_overlap(False)
    _, ref, _ = bench.hot_path_step(lib, the_map, subs, 0.6, 4.0, 60, groups[0])
    lib.set_overlap(True)
    for g in groups:
        bench.hot_path_step(lib, the_map, subs, 0.6, 4.0, 60, g)
    bad = [0]
    seen = [0]

    def check(tops):
        seen[0] += 1
        if not all(np.array_equal(a, b) for a, b in zip(tops, ref)):
            bad[0] += 1

    t0 = time.perf_counter()
    bench.run_steps(lib, the_map, subs, 0.6, 4.0, 60, groups, steps, after_step=check)
    lib.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %d steps, %d in flight, %.3f ms per step; steps compared %d, steps that differ from the serialised reference: %d"
          % (wl, steps, depth, 1e3 * dt / steps, seen[0], bad[0]))
    sys.exit(1 if bad[0] else 0)


if __name__ == "__main__":
    main()
