#!/usr/bin/env python3
"""Diagnostic (GPU box): statistics of the pose-scoring search on the bench workload -- how many transformed hi
points have any lo point nearby, at several radii.  Guides the design of k_pose_*; not part of the product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
import bench
from mad_amd import _lib
from mad_amd.eqsp import EQSP_Sphere
from mad_amd.orient_tables import orientation_matrices

W = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
lib.set_eqsp(1, e16.sphere_eqsp)
the_map, subs, _ = bench.build_inputs(lib, W, 0)
lo = lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index)
rng = np.random.default_rng(0)
for s in subs[:2]:
    hi = lib.set_build(s.slots, s.coords, s.octave, s.subv, s.index)
    top, idx, st = lib.match_topk(hi, lo, 0.6, 4.0, 60)
    n = st["n_pairs"]
    rows = lib.match_results(hi, lo, n)
    uh, ul = lib.match_used(len(s.coords), len(the_map.coords))
    hc, lc = np.unique(s.subv[uh], axis=0), np.unique(the_map.subv[ul], axis=0)
    print("pairs %d  hi cloud %d  lo cloud %d  lo bbox %s" % (n, len(hc), len(lc), np.round(lc.max(0) - lc.min(0), 1)))
    tree = cKDTree(lc)
    sel = rng.choice(n, size=min(n, 1500), replace=False)
    stats = {r: [] for r in (4.0, 5.0, 5.73, 6.6, 8.0)}
    cand = []
    cell = 8.02
    mn = lc.min(0)
    for p in sel:
        r = rows[p]
        R = r[14:23].reshape(3, 3)
        pts = (hc - r[8:11]) @ R.T + r[11:14]
        for rad in stats:
            d, _ = tree.query(pts, k=1, distance_upper_bound=rad)
            stats[rad].append(np.mean(np.isfinite(d)))
        # candidates the current kernel walks: lo points in the cells met by the ball of reach 4.01
        c0 = np.floor((pts - 4.01 - mn) / cell).astype(int)
        c1 = np.floor((pts + 4.01 - mn) / cell).astype(int)
        lcell = np.floor((lc - mn) / cell).astype(int)
        nc = np.array([np.count_nonzero(np.all((lcell >= a) & (lcell <= b), axis=1)) for a, b in zip(c0, c1)])
        cand.append(nc)
    for rad, v in stats.items():
        print("  any lo point within %.2f A: %.3f of the transformed points" % (rad, np.mean(v)))
    cand = np.concatenate(cand)
    print("  candidates per point: mean %.2f  P(0) %.3f  p90 %d  max %d ; mean of max over 64-point groups %.2f" % (
        cand.mean(), np.mean(cand == 0), np.percentile(cand, 90), cand.max(),
        np.mean([cand[i:i + 64].max() for i in range(0, len(cand) - 63, 64)])))
    print("  top-1 repeat %.1f ; mean count/l_hi over sample %.3f" % (top[0, 1], np.mean(stats[4.0])))
    hi.close()
