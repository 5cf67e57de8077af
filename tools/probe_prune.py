#!/usr/bin/env python3
"""How far the bounds of the pose search prune on the benchmark's workload: per subunit, the pairs that reach the exact
search, the match-count distribution and the counts of the k best pairs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402
from mad_amd import _lib      # noqa: E402
from mad_amd.eqsp import EQSP_Sphere      # noqa: E402
from mad_amd.orient_tables import orientation_matrices      # noqa: E402

lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
W = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
the_map, subs, _ = bench.build_inputs(lib, W, 0)
lo = lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index)
for sub in subs[:4]:
    hi = lib.set_build(sub.slots, sub.coords, sub.octave, sub.subv, sub.index)
    for k in (60, 840):
        top, idx, st = lib.match_topk(hi, lo, 0.6, 4.0, k)
        n_sel = lib.last_pose_selected()
        ph, pl, ps, cnt = lib.match_fetch(st["n_pairs"])
        order = np.lexsort((np.arange(len(cnt)), -cnt.astype(np.int64)))[:k]
        assert np.array_equal(order, idx)
        pct = np.percentile(cnt, [50, 90, 99, 99.9])
        print("k=%d pairs %d l_hi %d selected %d (%.1f %%)  count median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %d  k-th best %d"
              % (k, st["n_pairs"], st["l_hi"], n_sel, 100.0 * n_sel / st["n_pairs"], pct[0], pct[1], pct[2], pct[3], cnt.max(), cnt[order[-1]]))
    hi.close()
