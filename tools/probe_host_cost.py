#!/usr/bin/env python3
"""Host-side cost (enqueue time, nothing waited for) of the calls one rank of an 8-rank step makes: share build, export, import,
subunit build, match begin / finish.  C4 workload, rehearsal mode of ShardedSetBuild."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch      # noqa: E402,F401
torch.cuda.set_device(0)
import bench      # noqa: E402
from mad_amd import _lib, dist as mdist      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402
from mad_amd.orient_tables import orientation_matrices      # noqa: E402

lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
W = bench.WORKLOADS["c4"]
the_map, subs, _ = bench.build_inputs(lib, W, 0, 8)
b = mdist.ShardedSetBuild(lib, the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index, 0, 1, emulate=(8, 0))
hi = _lib.DeviceSet(lib)
b.resize()
sub = subs[0]
T = {}


def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0


N = 200
for it in range(N + 20):
    if it == 20:
        T.clear()
        lib.synchronize()
    t0 = time.perf_counter(); b._build_share(); tick("build_share (190 anchors)", t0)
    t0 = time.perf_counter(); lib.set_export(b.share, b.cap_rows, device_ptr=b.wire.data_ptr()); tick("export", t0)
    t0 = time.perf_counter()
    lo = lib.set_import(b.world, b.cap_rows, b.coords, b.octave, b.subv, b.index, device_ptr=b.gathered.data_ptr(), into=b.full)
    tick("import (1508 anchors)", t0)
    t0 = time.perf_counter(); lib.set_build(sub.slots, sub.coords, sub.octave, sub.subv, sub.index, into=hi); tick("build subunit (435 anchors)", t0)
    t0 = time.perf_counter(); h = lib.match_topk_many_begin([hi], lo, 0.6, 4.0, 60); tick("match begin", t0)
    t0 = time.perf_counter(); lib.match_topk_many_finish(h); tick("match finish (waits)", t0)
for k, v in T.items():
    print("%-32s %7.1f us per call" % (k, 1e6 * v / N))
