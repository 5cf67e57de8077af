#!/usr/bin/env python3
"""Where a row's workgroup of k_describe spends its time (diagnostic build with -DMAD_PROBE_STAMPS):
    MAD_LIB_PATH=mad_amd/csrc/build_stamps/libmad_amd_stamps.so python tools/probe_describe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402

NAMES = ["prologue (row -> anchor chain, stage tables, zero hist)", "indices + texel requests", "wait + classify + histogram", "queue reservation",
         "barrier", "queue phase (exact tiers)", "write-out + norm"]


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
    for what, src in (("map", the_map), ("subunit 0", subs[0])):
        s = _lib.DeviceSet(lib)
        for _ in range(3):
            lib.set_build(src.slots, src.coords, src.octave, src.subv, src.index, into=s)
        lib.synchronize()
        n = 16384
        out = np.zeros(n * 8, np.int64)
        assert lib.dll.mad_debug_dsc_stamps(out.ctypes.data_as(C.c_void_p), C.c_int(n * 8)) == 0
        st = out.reshape(n, 8).astype(np.float64)
        ok = (st[:, 7] > 0) & (st[:, 0] > 0)
        st = st[ok]
        st = st[np.argsort(st[:, 7])]      # the slots keep the stamps of earlier launches: take the last launch (no 40 us gap between its workgroups' ends)
        gaps = np.nonzero(np.diff(st[:, 7]) > 100000)[0]
        if len(gaps):
            st = st[gaps[-1] + 1:]
        d = np.diff(st, axis=1)
        print("%s: %d row workgroups; shader-clock ticks, median / p10 / p90" % (what, len(st)))
        for k, name in enumerate(NAMES):
            print("  %-58s %8.0f %8.0f %8.0f" % (name, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
        tot = st[:, 7] - st[:, 0]
        print("  workgroup total %.0f (median); kernel span %.0f ticks" % (np.median(tot), st[:, 7].max() - st[:, 0].min()))
        e = st[:, 7] - st[:, 7].min()
        print("  ends of the workgroups' last rows, ticks after the first one to end: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(e, q) for q in (10, 50, 90, 100)))


if __name__ == "__main__":
    main()
