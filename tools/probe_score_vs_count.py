#!/usr/bin/env python3
"""Does a pair's correlation score predict its match count?  For each subunit of the C3 workload: where, in the ranking by
score, the 60 pairs with the highest counts lie (the pose search could look at the best-scoring pairs first to raise its
pruning threshold early)."""
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
    top, idx, st = lib.match_topk(hi, lo, 0.6, 4.0, 60)
    ph, pl, ps, cnt = lib.match_fetch(st["n_pairs"])
    n = len(cnt)
    by_count = np.argsort(-cnt.astype(np.int64), kind="stable")[:60]
    rank_by_score = np.empty(n, np.int64)
    rank_by_score[np.argsort(-ps, kind="stable")] = np.arange(n)
    r = np.sort(rank_by_score[by_count]) / n
    # the k-th largest count among the best-scoring fraction f of the pairs, against the k-th largest overall
    kth = cnt[by_count[-1]]
    line = []
    for f in (0.01, 0.02, 0.05, 0.1, 0.25, 0.5):
        sel = np.argsort(-ps, kind="stable")[:max(int(f * n), 60)]
        c = np.sort(cnt[sel])[::-1]
        line.append("%.0f%%:%d" % (100 * f, c[59] if len(c) >= 60 else 0))
    print("pairs %d  k-th count %d  score rank of the top-60-by-count: median %.3f p90 %.3f max %.3f | 60th largest count within the best-scoring fraction: %s"
          % (n, kth, np.median(r), np.percentile(r, 90), r.max(), " ".join(line)))
    hi.close()
