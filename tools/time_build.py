import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from mad_amd import _lib
from mad_amd.eqsp import EQSP_Sphere
from mad_amd.orient_tables import orientation_matrices
lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj); lib.set_eqsp(1, e16.sphere_eqsp)
the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS["c3"])
lib.set_overlap(False)
for what, src in (("map", the_map), ("sub0", subs[0])):
    s = _lib.DeviceSet(lib)
    for _ in range(3): lib.set_build(src.slots, src.coords, src.octave, src.subv, src.index, into=s)
    lib.synchronize(); lib.timers_reset() if hasattr(lib, "timers_reset") else None
    t0 = time.perf_counter()
    for _ in range(20): lib.set_build(src.slots, src.coords, src.octave, src.subv, src.index, into=s)
    lib.synchronize()
    print(what, "build ms", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
