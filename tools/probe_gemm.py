#!/usr/bin/env python3
"""Where a tile of k_corr_gemm2 spends its time (diagnostic build with -DMAD_PROBE_STAMPS, s_memtime at the phases of every
workgroup's first tile):   MAD_LIB_PATH=mad_amd/csrc/build_stamps/libmad_amd_stamps.so python tools/probe_gemm.py"""
import ctypes as C
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
    W = bench.WORKLOADS["c3"]
    the_map, subs, _ = bench.build_inputs(lib, W)
    sets = [_lib.DeviceSet(lib) for _ in range(1 + len(subs))]
    lib.set_overlap(False)
    lo, his = bench.enqueue_builds(lib, the_map, subs, sets)
    for _ in range(3):
        lib.match_topk(his[0], lo, 0.6, 4.0, 60)
    lib.synchronize()
    n = 512
    out = np.zeros(n * 8, np.int64)
    rc = lib.dll.mad_debug_g2_stamps(out.ctypes.data_as(C.c_void_p), C.c_int(n * 8))
    assert rc == 0
    st = out.reshape(n, 8).astype(np.float64)
    t0 = st[:, 0].min()
    names = ["kernel start -> first tile started (norms, pieces of two stages issued)", "accumulators", "first stage", "K loop (15 stages)",
             "barrier + start of the NEXT tile", "epilogue"]
    d = np.diff(st[:, :7], axis=1)
    print("s_memtime ticks (100 MHz constant clock on gfx9: 1 tick = 10 ns); median / p10 / p90 over %d workgroups" % n)
    for k, name in enumerate(names):
        print("  %-22s %8.0f %8.0f %8.0f" % (name, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
    print("  first tile total       %8.0f ;  start spread %8.0f ; last end - first start %8.0f" % (
        np.median(st[:, 6] - st[:, 0]), st[:, 0].max() - t0, st[:, 6].max() - t0))


if __name__ == "__main__":
    main()
