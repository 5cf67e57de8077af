#!/usr/bin/env python3
"""Where a workgroup of k_describe_ball spends its time (diagnostic build with -DMAD_PROBE_STAMPS):
    tools/build_variant.sh stamps -DMAD_PROBE_STAMPS -fno-slp-vectorize -fno-vectorize
    MAD_LIB_PATH=mad_amd/csrc/build_stamps/libmad_amd_stamps.so python tools/probe_ball.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402

NAMES = ["prologue: tables + ball into LDS, barriers", "table tier (wave 0: 16 samples per thread)", "barrier", "open samples (exact tiers)",
         "barrier", "-", "write-out"]


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
    n = 16384
    out = np.zeros(n * 8, np.int64)
    rows = np.zeros(n, np.int32)
    for what, src in (("map", the_map), ("subunit 0", subs[0])):
        s = _lib.DeviceSet(lib)
        for _ in range(3):
            lib.set_build(src.slots, src.coords, src.octave, src.subv, src.index, into=s)
            lib.synchronize()
            used = lib.dll.mad_debug_dscb_stamps(out.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), C.c_int(n))
        assert used > 0, used
        st = out.reshape(n, 8)[:used].astype(np.float64)
        nr = rows[:used]
        print("%s: %d workgroups with rows (%s by rows 1..4); shader-clock ticks, median / p10 / p90" % (what, used, np.bincount(nr, minlength=5)[1:].tolist()))
        full = nr >= 1
        d = np.diff(st[full], axis=1)
        for k, name in enumerate(NAMES):
            print("  %-58s %8.0f %8.0f %8.0f" % (name, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
        tot = st[:, 7] - st[:, 0]
        for r in range(1, 5):
            if (nr == r).any():
                print("  workgroups of %d rows: total %.0f (median)" % (r, np.median(tot[nr == r])))
        real = np.zeros(n * 2, np.int64)
        assert lib.dll.mad_debug_dscb_real(real.ctypes.data_as(C.c_void_p), C.c_int(n)) == 0
        real = real.reshape(n, 2)[:used].astype(np.float64) * 0.01      # us
        t0 = real[:, 0].min()
        print("  device clock: workgroups start %.1f / %.1f / %.1f / %.1f us (p10 / p50 / p90 / max) after the first, live %.1f / %.1f us (p50 / p90), last ends at %.1f us" % (
            tuple(np.percentile(real[:, 0] - t0, q) for q in (10, 50, 90, 100)) + tuple(np.percentile(real[:, 1] - real[:, 0], q) for q in (50, 90)) + (real[:, 1].max() - t0,)))
        live = real[:, 1] - real[:, 0]
        dbg = np.zeros(n * 4, np.int32)
        assert lib.dll.mad_debug_dscb_dbg(dbg.ctypes.data_as(C.c_void_p), C.c_int(n)) == 0
        dbg = dbg.reshape(n, 4)[:used]
        print("  open samples of a workgroup's fullest row: p50 %d p90 %d p99 %d max %d; per workgroup: open %.0f, float64 voxels %.1f, float64 tier %.1f (means)" % (
            np.median(dbg[:, 0]), np.percentile(dbg[:, 0], 90), np.percentile(dbg[:, 0], 99), dbg[:, 0].max(), dbg[:, 1].mean(), dbg[:, 2].mean(), dbg[:, 3].mean()))
        for w in np.argsort(-live)[:6]:
            print("    long-lived workgroup: %d rows, start %.1f us, live %.1f us, phases (ticks) %s, dbg [fullest row, open, f64 voxels, f64 tier] %s" % (nr[w], real[w, 0] - t0, live[w], np.diff(st[w]).astype(np.int64).tolist(), dbg[w].tolist()))
        order = np.argsort(real[:, 0])
        print("  starts (us) of every 16th workgroup in start order:", np.round(real[order, 0][::16] - t0, 1).tolist()[:40])
        print("  kernel span %.0f ticks; starts p50 %.0f p90 %.0f max %.0f after the first" % ((st[:, 7].max() - st[:, 0].min(),) + tuple(np.percentile(st[:, 0] - st[:, 0].min(), q) for q in (50, 90, 100))))


if __name__ == "__main__":
    main()
