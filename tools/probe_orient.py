#!/usr/bin/env python3
"""Where an anchor's workgroup of k_orient spends its time (diagnostic build with -DMAD_PROBE_STAMPS):
    MAD_LIB_PATH=mad_amd/csrc/build_stamps/libmad_amd_stamps.so python tools/probe_orient.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402

NAMES = ["init (zero hist, stage tables)", "fetch + compact voxels", "first binning", "exact queue 1", "quantise + main bins (wave 0)",
         "stage rotations", "re-binning per candidate", "exact queue 2", "quantise + secondary (wave per cand.)", "emit"]


def main():
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    lib = _lib.Lib(0)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    the_map, subs, _ = bench.build_inputs(lib, bench.WORKLOADS["c3"])
    lib.set_overlap(False)
    s = _lib.DeviceSet(lib)
    for _ in range(3):
        lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index, into=s)
    lib.synchronize()
    n = min(len(the_map.coords), 4096)
    out = np.zeros(n * 12, np.int64)
    assert lib.dll.mad_debug_ori_stamps(out.ctypes.data_as(C.c_void_p), C.c_int(n * 12)) == 0
    st = out.reshape(n, 12)[:, :11].astype(np.float64)
    ok = st[:, 10] > 0      # anchors rejected early have no end stamp
    st = st[ok]
    d = np.diff(st, axis=1)
    print("%d anchors with rows; shader-clock ticks, median / p10 / p90" % len(st))
    for k, name in enumerate(NAMES):
        print("  %-40s %8.0f %8.0f %8.0f" % (name, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
    tot = st[:, 10] - st[:, 0]
    print("  workgroup total %.0f (median); kernel span %.0f ticks" % (np.median(tot), st[:, 10].max() - st[:, 0].min()))


if __name__ == "__main__":
    main()
