"""Diagnostic: does the overlapped (multi-lane) hot path reproduce the serialised one bit for bit?

  WL=c3|c5 ITERS=n MODE=all|sync_between python tools/probe_overlap.py      (on the GPU box)

Builds every set of the benchmark workload serially as the reference, then repeats builds + matches with the lanes
overlapping and reports every set whose descriptors, and every match whose top-k, differ from the reference.  This
is the script that exposed the packed-float32 hazard recorded in DESIGN.md section 5b."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mad_amd import _lib
from mad_amd.eqsp import EQSP_Sphere
from mad_amd.orient_tables import orientation_matrices
lib = _lib.Lib(0)
e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
dom, adj = orientation_matrices(e112)
lib.set_eqsp(0, e112.sphere_eqsp, dom, adj); lib.set_eqsp(1, e16.sphere_eqsp)
W = bench.WORKLOADS[os.environ.get("WL", "c5")]
the_map, subs, _ = bench.build_inputs(lib, W, 0)
structs = [the_map] + subs
sets = [_lib.DeviceSet(lib) for _ in structs]
mode = os.environ.get("MODE", "all")
# reference descriptors: serial builds
lib.set_overlap(False)
for st, d in zip(structs, sets):
    lib.set_build(st.slots, st.coords, st.octave, st.subv, st.index, into=d)
lib.synchronize()
ref = [s.download() for s in sets]
ref_tops = [(t.copy(), i.copy()) for t, i, _ in lib.match_topk_many(sets[1:], sets[0], 0.6, 4.0, 60)]
lib.set_overlap(True)
for it in range(int(os.environ.get("ITERS", "4"))):
    for st, d in zip(structs, sets):
        lib.set_build(st.slots, st.coords, st.octave, st.subv, st.index, into=d)
    if mode == "sync_between":
        lib.synchronize()
    res = lib.match_topk_many(sets[1:], sets[0], 0.6, 4.0, 60)
    for mi, ((t, i, _), (rt, ri)) in enumerate(zip(res, ref_tops)):
        if not (np.array_equal(t, rt) and np.array_equal(i, ri)):
            print("iter", it, "match", mi, "top-k differs from the serial run")
    rows = [s.download() for s in sets]
    for si, (a, b) in enumerate(zip(ref, rows)):
        if not np.array_equal(a["dsc"], b["dsc"]):
            bad = np.argwhere(a["dsc"] != b["dsc"])
            rws = sorted(set(bad[:, 0].tolist()))
            print("iter", it, "set", si, "lane", si % 8, "dsc wrong: rows", rws[:6], "n", len(bad), "of rows", len(a["dsc"]))
            r0 = rws[0]
            d = b["dsc"][r0].astype(int) - a["dsc"][r0].astype(int)
            nz = np.flatnonzero(d)
            print("   row", r0, "sum ref", a["dsc"][r0].sum(), "got", b["dsc"][r0].sum(), "diff bins", nz[:12], "deltas", d[nz][:12], "subcubes", sorted(set((nz // 16).tolist()))[:16])
            r1 = rws[-1]
            d = b["dsc"][r1].astype(int) - a["dsc"][r1].astype(int)
            nz = np.flatnonzero(d)
            print("   row", r1, "sum ref", a["dsc"][r1].sum(), "got", b["dsc"][r1].sum(), "n diff", len(nz), "deltas", d[nz][:12])
print("done", mode)
lib.synchronize()
lib.set_overlap(False)
for st, d in zip(structs, sets):
    lib.set_build(st.slots, st.coords, st.octave, st.subv, st.index, into=d)
lib.synchronize()
again = [s.download() for s in sets]
nbad = sum(0 if np.array_equal(a["dsc"], b["dsc"]) else 1 for a, b in zip(ref, again))
print("serial rebuild after the overlapped iterations: sets differing from ref:", nbad)
