#!/usr/bin/env python3
"""Times the build stage alone (orient + describe of all structures of a step, one batch), serialised, with HIP events:
    MAD_LIB_PATH=<variant .so> python tools/probe_build.py [workload] [reps]
For diagnostic builds whose descriptors are not meant to be right (texel-size probes): nothing downstream is run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    lib = _lib.Lib(0)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS[wl])
    sets = [_lib.DeviceSet(lib) for _ in range(1 + len(subs))]
    lib.set_overlap(False)
    for _ in range(3):
        bench.enqueue_builds(lib, the_map, subs, sets)
    rows = [s.size()[0] for s in sets]
    lib.timing_enable(True)
    lib.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        bench.enqueue_builds(lib, the_map, subs, sets)
    lib.synchronize()
    wall = (time.perf_counter() - t0) / reps
    out = {g: lib.timing_get(g)[0] / reps for g in ("orient", "describe")}
    print("lib=%s workload=%s rows=%d orient %.4f ms describe %.4f ms wall %.4f ms per step" % (
        os.path.basename(_lib.LIB_PATH), wl, sum(rows), out["orient"], out["describe"], wall * 1e3))


if __name__ == "__main__":
    main()
