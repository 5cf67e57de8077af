#!/usr/bin/env python3
"""bench.py -- throughput of the MaD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c4|c2|c5|small]

One "step" = one pass of the hot path over one batch of synthetic input already resident
in HBM: orientation + description of the map and of every subunit (mad_set_build),
then, per subunit, int8-MFMA correlation, pair compaction, pose scoring and top-k
(mad_match_topk).  Metric = (sum over subunits of N_hi_rows x N_lo_rows) / wall time,
"anchor-pair x rotation correlations/s" (BASELINE.json; unit definition SURVEY.md 8(d)).

N = 1 defaults to C3 (BASELINE configs[2]: 256^3 map, 4 subunits), the configuration the metric is quoted on.
N > 1 defaults to C4 (configs[3]: 256^3 map, 8 subunits, seeds 30-37) with STRONG scaling: the same total work at every
N.  One process per GPU (`python bench.py --gpus N` starts them itself, before anything touches a GPU; under
torch.distributed.run it joins the ranks it is given):
  * stage A -- the map's anchors are dealt round-robin to the ranks, each rank orients + describes its share, ONE RCCL
    all-gather of the rows (int8 descriptors, norms, bin ids) gives every rank the full map set (mad_amd/dist.py,
    ShardedSetBuild);
  * the subunits are dealt round-robin to the ranks; each rank builds and matches its own against the full map set
    (no data-path collective: the pair grid of a subunit is independent of the others);
  * ONE all-gather of the per-subunit top-k poses per step, collected a step later (overlaps the next step's kernels).
value = correlations of ALL ranks / max-over-ranks time.  `--workload c4 --gpus 1` is the one-GPU point of that curve.

Inputs are synthetic (seeded pseudo-atom assemblies).  The density grids
come from this library's own GPU density simulation, the scale space and the anchors from
its device MapSpace / Detector (upstream of the hot path, untimed setup, times reported
in config.setup_detail_s).

The JSON line also carries `roofline` (dominant kernel, HIP-event timed on the library's
stream inside the timed region) and `cpu_baseline` (the CPU oracle timed on a bounded
sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def load_workloads():
    """The workloads of SURVEY.md 8(d), frozen in bench/configs/*.json (seeds, atom counts, radii, lattice, noise): one file per
    BASELINE.json config, plus c3clean (C3 without its noise: the workload rounds 1-2 were measured on)."""
    out = {}
    d = os.path.join(ROOT, "bench", "configs")
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".json"):
            with open(os.path.join(d, fn)) as fh:
                c = json.load(fh)
            c["n_sub"] = len(c["seeds"])                 # distinct subunits (sets matched against the map)
            c["n_placed"] = c["n_sub"] * c["copies"]     # rigid copies in the map
            c["grid"] = tuple(c["lattice"])
            c["file"] = "bench/configs/" + fn
            out[fn[:-5]] = c
    out["small"] = out["c1"]
    return out


WORKLOADS = load_workloads()
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_PEAK_TOPS = 5000.0      # dense int8 MFMA = 2x bf16 (~2.5 PF)
ORIENT_BYTES = 58956       # 17^3 x 3 x f32 read per anchor
DESCRIBE_BYTES = 51200     # 4096 x 12 B gathered + 2048 B written per row
REFINE_BYTES = 96          # 8 corners x 12 B gathered per atom per refinement step
CCC_BYTES = 8              # two float32 grids read per voxel of the overlap box
REFERENCE_PY_CORR_S = 2.0e4      # SURVEY.md section 6 [probe]: the reference itself (python + numpy, one thread), 1.5e5 correlations in 6.9 s


def source_fingerprint():
    """sha256 over the kernel sources: ties a committed PMC summary (profiles/) to the build it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mad_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) or fn == "Makefile":
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fn.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------------------------
# untimed setup: synthetic structures, their scale space and anchors (the library's own upstream stages)
# --------------------------------------------------------------------------------------------

SETUP_T = {}


class Structure(object):
    """Device-resident fields + anchors of one structure: density simulation (a14-a15), MapSpace and Detector
     all through the library; nothing of it is inside the timed region."""

    def __init__(self, lib, atoms, mass, res, vs, N=None, tag="sub", noise=0.0):
        from mad_amd.Detector import Detector
        from mad_amd.MapSpace import MapSpace
        grid, x0, y0, z0 = lib.structure_to_density(atoms, mass, res, vs)
        origin = np.array([x0, y0, z0])
        if N is not None:      # centre the simulated density in an exactly N^3 box
            assert max(grid.shape) <= N, "assembly of %s voxels does not fit %d^3" % (grid.shape, N)
            lo = [(N - s) // 2 for s in grid.shape]
            big = np.zeros((N, N, N), np.float32)
            big[lo[0]:lo[0] + grid.shape[0], lo[1]:lo[1] + grid.shape[1], lo[2]:lo[2] + grid.shape[2]] = grid
            origin = origin - np.array(lo) * vs
            grid = big
        if noise > 0.0:      # Gaussian noise sigma_n x max over the whole box, seeded
            grid = (grid + np.random.default_rng(4321).normal(0.0, noise * float(grid.max()), grid.shape)).astype(np.float32)
        self.shape = grid.shape
        self.grid, self.origin, self.atoms, self.mass = grid, origin, atoms, mass
        ms = MapSpace(tag + ".pdb", resolution=res, voxelsp=vs)
        lib.synchronize()
        t0 = time.perf_counter()
        ms.build_from_grid(grid, float(origin[0]), float(origin[1]), float(origin[2]), lib=lib)
        lib.synchronize()
        SETUP_T[tag + "_mapspace_s"] = SETUP_T.get(tag + "_mapspace_s", 0.0) + time.perf_counter() - t0
        t0 = time.perf_counter()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            anchors = Detector().find_anchors(ms)
        SETUP_T[tag + "_detector_s"] = SETUP_T.get(tag + "_detector_s", 0.0) + time.perf_counter() - t0
        self.ms = ms
        self.slots = ms.device_slots(lib)
        self.coords = np.array([a.coords for a in anchors], np.int32).reshape(-1, 3)
        self.octave = np.array([a.oct_scale for a in anchors], np.int32)
        self.subv = np.array([a.subv_map_coords for a in anchors], np.float64).reshape(-1, 3)
        self.index = np.arange(len(anchors), dtype=np.int32)
        self.item = 0      # position of a subunit in the workload's list

    def anchor_list(self, variant=0):
        """The structure's anchors as mad_set_build takes them.  variant 1: the same anchors listed in reverse (FRESH): a set
        rebuilt with the other list cannot take the library's unchanged-anchors shortcut, so the step pays for the whole upload."""
        if variant == 0:
            return self.slots, self.coords, self.octave, self.subv, self.index
        rev = self.__dict__.get("_rev")
        if rev is None:
            rev = self._rev = tuple(np.ascontiguousarray(a[::-1]) for a in (self.coords, self.octave, self.subv, self.index))
        return (self.slots,) + rev

    def base_gradient(self):
        """(3, X, Y, Z) float32 gradient of the base octave on the host, for the CPU baseline sample."""
        return self.gradient(1)

    def gradient(self, octave):
        """(3, X, Y, Z) float32 gradient of one octave on the host (what the CPU oracle samples); kept after the first call."""
        cache = self.__dict__.setdefault("_g", {})
        if octave not in cache:
            cache[octave] = np.ascontiguousarray(np.moveaxis(self.ms.grad_list[octave], -1, 0), dtype=np.float32)
        return cache[octave]


def build_inputs(lib, W, rank=0, world=1, items=None):
    """-> (map, this rank's subunits, setup seconds).  The map (every subunit placed `copies` times on a jittered lattice, plus
    the workload's Gaussian noise) is the same on every rank; subunit s belongs to rank s % world (strong scaling: the workload
    does not grow with the ranks), or -- `items` -- the rank takes exactly the listed subunits, in that order (dist.plan_partition)."""
    from mad_amd import synth
    rng = np.random.default_rng(1234)
    subs, placed, placed_mass = [], [], []
    sp = 2.2 * W["radius"]
    cells = [(i, j, k) for i in range(W["grid"][0]) for j in range(W["grid"][1]) for k in range(W["grid"][2])]
    assert len(cells) >= W["n_placed"], "lattice of %d cells for %d copies" % (len(cells), W["n_placed"])
    centre = (np.array(W["grid"]) - 1) * sp / 2
    for s, seed in enumerate(W["seeds"]):
        atoms, names, elems = synth.random_globule(W["n_atoms"], W["radius"], seed=seed)
        for c in range(W["copies"]):
            placed.append(synth.place(atoms, synth.random_rotation(rng), np.array(cells[s * W["copies"] + c]) * sp - centre + rng.normal(scale=2.0, size=3)))
            placed_mass.append(synth.masses(elems))
        if (s % world == rank) if items is None else (s in items):
            subs.append((s, atoms, synth.masses(elems)))
    if items is not None:
        subs.sort(key=lambda t: list(items).index(t[0]))
    mass_all = np.concatenate(placed_mass)
    t0 = time.time()
    the_map = Structure(lib, np.concatenate(placed), mass_all, W["res"], W["vs"], N=W["N"], tag="map", noise=W.get("noise", 0.0))
    the_map.placed = placed      # the copies as they sit in the map: subunit s, copy c = placed[s * copies + c]
    sub_structs = []
    for s, a, m in subs:
        st = Structure(lib, a, m, W["res"], W["vs"])
        st.item = s
        sub_structs.append(st)
    return the_map, sub_structs, time.time() - t0


# --------------------------------------------------------------------------------------------
# the timed step
# --------------------------------------------------------------------------------------------

HOST_T = {}


_BATCHES = {}
BATCHED = {"on": os.environ.get("MAD_BUILD_BATCH", "0") == "1", "asked": "MAD_BUILD_BATCH" in os.environ}
# FRESH: every rebuild of a group of sets gets the OTHER listing of the same anchors (Structure.anchor_list), so that the library's
# upload does its whole work in every timed step -- duplicate canonicalisation, Morton sort, 90 KB of staging per step -- instead of
# recognising the lists of the previous step (set_upload_anchors' `same` branch).  Off: the lists repeat, as in rounds 1-3.
FRESH = {"on": False, "uses": {}}


def enqueue_builds(lib, the_map, subs, sets):
    """orient + describe of the map and of every subunit into `sets` (device-resident, rebuilt in place); asynchronous.
    sets[0] is the map's DeviceSet, or a dist.ShardedSetBuild when the map's rows are built in shares over the ranks.
    One `mad_set_build` per structure, each on its own lane (the default: the launches of different structures overlap and every
    match starts as soon as its own sets are ready), or -- BATCHED (`--batched`, MAD_BUILD_BATCH=1) -- all structures of the step
    in one `mad_set_build_many` (one launch per stage: less device time, but one long dependency chain per step)."""
    t0 = time.perf_counter()
    shared = hasattr(sets[0], "enqueue")
    var = 0
    if FRESH["on"]:
        var = FRESH["uses"].get(id(sets), 0) & 1
        FRESH["uses"][id(sets)] = var + 1
    if not BATCHED["on"]:
        if shared:
            lo = sets[0].enqueue()
        else:
            lo = lib.set_build(*the_map.anchor_list(var), into=sets[0])
        his = [lib.set_build(*s.anchor_list(var), into=d) for s, d in zip(subs, sets[1:])]
    else:
        ent = _BATCHES.get(id(sets))
        if ent is None or ent[1] is not sets:
            first = sets[0].build_job() if shared else (the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index, sets[0])
            jobs = [first] + [(s.slots, s.coords, s.octave, s.subv, s.index, d) for s, d in zip(subs, sets[1:])]
            # MAD_BUILD_GROUPS batches, each on the lane of its first set (they overlap); the structures are dealt largest first
            # to the batch with the fewest anchors so far
            n_groups = max(1, min(int(os.environ.get("MAD_BUILD_GROUPS", "1")), len(jobs)))
            groups = [[] for _ in range(n_groups)]
            for i in sorted(range(len(jobs)), key=lambda i: -len(jobs[i][2])):
                min(groups, key=lambda g: sum(len(jobs[j][2]) for j in g)).append(i)
            groups = [sorted(g) for g in groups if g]
            ent = _BATCHES[id(sets)] = ([(lib.prepare_build_many([jobs[i] for i in g]), g) for g in groups], sets)      # made once per group of sets
        built = [None] * len(sets)
        for batch, g in ent[0]:
            for i, d in zip(g, batch.run()):
                built[i] = d
        lo = sets[0].finish() if shared else built[0]
        his = built[1:]
    HOST_T["build_enqueue"] = HOST_T.get("build_enqueue", 0.0) + time.perf_counter() - t0
    return lo, his


PM = {"pm": None}      # dist.PartitionedMatch when some subunits' pair grids are split over groups of ranks (n_sub % ranks != 0)


# A map set built in shares travels in wire images of a fixed capacity (dist.ShardedSetBuild).  A build that outgrows it makes the
# import report MAD_ENOSPC at the first use of the set -- on every rank that uses it, the images being the same everywhere.  The rank
# notes it here, hands its word to the others with the step's top-k exchange (a rank without a match in that step learns it there),
# and all ranks then size the images again at the same step (main: ResizeMapImages).
OVERFLOW = {"seen": 0, "resizes": 0, "watch": False}


class ResizeMapImages(Exception):
    pass


def _overflowed(err):
    if OVERFLOW["watch"] and "ENOSPC" in str(err) and "mad_set_import" in str(err):
        OVERFLOW["seen"] = 1
        return True
    return False


def begin_matches(lib, his, lo, cc, dist, k):
    try:
        if PM["pm"] is not None:
            return PM["pm"].begin(lib, his, lo, cc, dist, k)
        return lib.match_topk_many_begin(his, lo, cc, dist, k)
    except RuntimeError as e:      # _lib.MadBackendError
        if not _overflowed(e):
            raise
        return None


def collect(lib, handle, his):
    t0 = time.perf_counter()
    try:
        if handle is None:      # the step's matches were not begun (begin_matches)
            corr, tops, stats = 0, [], []
        elif PM["pm"] is not None:
            corr, tops, stats = PM["pm"].finish(lib, handle)
        else:
            corr, tops, stats = 0, [], []
            for top, idx, st in lib.match_topk_many_finish(handle):
                corr += st["n_corr"]
                tops.append(top)
                stats.append(st)
    except RuntimeError as e:      # _lib.MadBackendError
        if not _overflowed(e):
            raise
        corr, tops, stats = 0, [], []
    HOST_T["match_wait"] = HOST_T.get("match_wait", 0.0) + time.perf_counter() - t0
    for hi, st in zip(his, stats):
        if "n_hi" not in st:
            st["n_hi"], _ = hi.size()
            st["n_lo"] = st["n_corr"] // max(st["n_hi"], 1)
    return corr, tops, stats


def hot_path_step(lib, the_map, subs, cc, dist, k, sets):
    """One step, start to finish: returns (correlations, [top-k result rows per subunit], stats).  Everything is enqueued
    asynchronously; the only host round trip is the result read-back that ends each match."""
    lo, his = enqueue_builds(lib, the_map, subs, sets)
    return collect(lib, begin_matches(lib, his, lo, cc, dist, k), his)


def run_steps(lib, the_map, subs, cc, dist, k, set_groups, n_steps, after_step=None):
    """n_steps steps with len(set_groups) of them in flight (2 to 4): while the matches of step i run, the host already
    enqueues the builds of step i + 1 into the next group of device sets and, with three (four) groups, the matches of step i are
    only collected after those of step i + 1 (and i + 2) have been enqueued (up to three open match brackets), so the device does
    not idle over the host's turn-around between steps.  Every step does the same work as hot_path_step; `after_step(tops)` is called once
    per step, in order."""
    depth = len(set_groups)
    lag = depth - 2      # how many steps behind the enqueue front the results are collected
    out, open_steps = None, []
    built = enqueue_builds(lib, the_map, subs, set_groups[0]) if n_steps > 0 else None

    def finish_oldest():
        handle, his = open_steps.pop(0)
        res = collect(lib, handle, his)
        if after_step is not None:
            after_step(res[1])
        return res

    try:
        for i in range(n_steps):
            lo, his = built
            t0 = time.perf_counter()
            open_steps.append((begin_matches(lib, his, lo, cc, dist, k), his))
            HOST_T["match_enqueue"] = HOST_T.get("match_enqueue", 0.0) + time.perf_counter() - t0
            if i + 1 < n_steps:      # its group was last read by step i + 1 - depth, collected by now
                built = enqueue_builds(lib, the_map, subs, set_groups[(i + 1) % depth])
            if len(open_steps) > lag:
                out = finish_oldest()
        while open_steps:
            out = finish_oldest()
    except ResizeMapImages:      # raised by after_step on every rank at the same step: what is in flight is collected and dropped
        while open_steps:
            handle, his = open_steps.pop(0)
            collect(lib, handle, his)
        raise
    return out


def refine_ccc_leg(lib, the_map, subs, tops, W, n_cand=8, reps=3):
    """refinement + CCC reported as their own line (candidates/s).  The n_cand best poses of
    every subunit are refined against the map (a13, all candidates in one launch), then each refined copy is turned
    into a simulated density (a14-a15) and scored by CCC against the map (a16) without leaving the device."""
    lib.upload_density(the_map.grid, the_map.origin, W["vs"])
    n_done, n_conv, best, steps_total, vox = 0, 0, [], 0, 0
    # ONE device call for all subunits (mad_dock_refine_score): the poses of the n_cand best rows of each go in (MaD.py:451: hi point,
    # lo point, rotation), the atoms are placed, refined, turned into densities and scored on the device; the coordinates stay there
    hi_p, lo_p, rot, owner = [], [], [], []
    for si, (sub, top) in enumerate(zip(subs, tops)):
        m = min(n_cand, len(top))
        if m == 0:
            continue
        hi_p.append(top[:m, 8:11]); lo_p.append(top[:m, 11:14])
        rot.append(np.swapaxes(top[:m, 14:23].reshape(m, 3, 3), 1, 2).reshape(m, 9))      # rows hold x' = R (x - hi) + lo: rotate_atoms takes R^T
        owner += [si] * m
        vox += m * int(np.prod(sub.shape))
    dts = []
    for _ in range(reps if owner else 0):      # the same call `reps` times, the median reported: the first one also sizes the library's buffers
        lib.timing_reset()
        lib.synchronize()
        t0 = time.perf_counter()
        _, conv, last, ccc = lib.dock_refine_score([s_.atoms for s_ in subs], [s_.mass for s_ in subs], np.concatenate(hi_p), np.concatenate(lo_p),
                                                   np.concatenate(rot), W["res"], want_coords=False, cand_struct=np.array(owner, np.int32))
        lib.synchronize()
        dts.append(time.perf_counter() - t0)
    if owner:
        n_done, n_conv = len(owner), int(np.sum(conv))
        steps_total = int(np.sum(np.asarray(last) + 1))
        best = [float(np.max(ccc[np.array(owner) == si])) for si in sorted(set(owner))]
    dt = float(np.median(dts)) if dts else 1.0
    ms = {g: lib.timing_get(g)[0] for g in ("refine", "density", "ccc")}
    n_atoms = int(len(subs[0].atoms)) if subs else 0
    roofs = {}
    if ms["refine"] > 0:      # trilinear gather of the map gradient at every atom, every step (structure_utils.py:106)
        a = REFINE_BYTES * n_atoms * steps_total / (ms["refine"] * 1e-3) / 1e9
        roofs["refine"] = dict(kernel="k_refine", bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS, refine_steps=steps_total,
                               note="latency-bound by construction: <= 500 dependent steps per candidate, the gathers of one step are L2-resident")
    if ms["ccc"] > 0 and vox:
        a = CCC_BYTES * vox / (ms["ccc"] * 1e-3) / 1e9
        roofs["ccc"] = dict(kernel="k_ccc_b", bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS, voxels=vox)
    if ms["density"] > 0 and vox:      # splat + three separable blur passes (float64 read + write each) + float32 conversion
        a = (3 * 16 + 12) * vox / (ms["density"] * 1e-3) / 1e9
        roofs["density"] = dict(kernel="k_splat_b + k_blur_b x3 + k_to_f32_b + k_norm_b (all candidates per launch)", bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS)
    return dict(value=n_done / dt, unit="candidates/s", candidates=n_done, converged=n_conv, seconds=dt, seconds_each=[round(x, 5) for x in dts], atoms_per_candidate=n_atoms,
                best_ccc_per_subunit=[round(float(b), 4) for b in best], kernel_ms=ms, roofline=roofs,
                note="mad_dock_refine_score, one call for the candidates of all subunits: placement, refinement, density simulation and CCC on the "
                     "device, poses in and scores out (the refined coordinates are not fetched); not part of the headline metric")


def cpu_baseline(the_map, subs, cc, dist, k, lib, n_lo_anchor=800, n_hi_anchor=250, threads=1, whole=False):
    """The CPU oracle on the same workload.  whole=False: a bounded sample, the base-octave anchors of the map and of every
    subunit (the map described once, every subunit docked into it: ~10 s on one core).  whole=True: EVERY anchor of both
    octaves -- the full step, which is also what the top-k agreement of the line is checked on.  threads > 1: the same scalar C
    functions on contiguous chunks of the work in that many host threads (oracle.py, *_mt)."""
    from mad_amd.eqsp import EQSP_Sphere
    from oracle import oracle as O
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)

    def described(st, n):
        parts = []
        for octave in ((0, 1) if whole else (1,)):
            sel = np.flatnonzero(st.octave == octave)
            if not whole:
                sel = sel[:n]
            if not len(sel):
                continue
            g = st.gradient(octave)
            rows = O.orient_mt(g[0], g[1], g[2], octave, st.coords[sel], e112.sphere_eqsp, e112.p_centers_eqsp, threads, want_counts=False)
            dsc = O.describe_mt(g[0], g[1], g[2], octave, st.coords[sel][rows["anchor"]], rows["R"], e16.sphere_eqsp, threads)
            parts.append((sel, rows, dsc, octave))
        sel_all = np.concatenate([p[0] for p in parts])
        first = np.cumsum([0] + [len(p[0]) for p in parts])
        anchor = np.concatenate([f + p[1]["anchor"] for f, p in zip(first, parts)])      # index into sel_all; the Detector lists octave 0 first
        return dict(sel=sel_all, anchor=anchor, R=np.concatenate([p[1]["R"] for p in parts]), main=np.concatenate([p[1]["main"] for p in parts]),
                    dsc=np.concatenate([p[2] for p in parts]), octave=np.concatenate([np.full(len(p[1]["anchor"]), p[3]) for p in parts]),
                    coords=st.coords[sel_all], subv=st.subv[sel_all], anc_octave=st.octave[sel_all])

    for st in [the_map] + list(subs):      # host copies of the fields (a download + np.gradient): not part of the CPU's time
        for o in ((0, 1) if whole else (1,)):
            st.gradient(o)
    t0 = time.perf_counter()
    lo_s = described(the_map, n_lo_anchor)
    lo_p = lo_s["subv"][lo_s["anchor"]]
    meta_l = np.stack([lo_s["anchor"], lo_s["octave"], lo_s["main"]], 1)
    his, n_corr, n_pairs = [], 0, 0
    for sub in subs:
        hi_s = described(sub, n_hi_anchor)
        ph, pl, ps, _ = O.correlate_mt(hi_s["dsc"], lo_s["dsc"], cc, threads)
        hi_s["pairs"], hi_s["order"], hi_s["res"] = len(ph), np.zeros(0, np.int64), None
        if len(ph):
            hi_p = hi_s["subv"][hi_s["anchor"]]
            meta_h = np.stack([hi_s["anchor"], hi_s["octave"], hi_s["main"]], 1)
            res, cnt = O.pose_score_mt(ph, pl, ps, hi_p, hi_s["R"], meta_h, lo_p, lo_s["R"], meta_l,
                                       np.unique(hi_p[np.unique(ph)], axis=0), np.unique(lo_p[np.unique(pl)], axis=0), dist, threads)
            hi_s["order"], hi_s["res"] = O.topk(cnt, k), res
        n_corr += len(hi_s["dsc"]) * len(lo_s["dsc"])
        n_pairs += len(ph)
        his.append(hi_s)
    dt = time.perf_counter() - t0
    # the same anchors through the GPU path: top-k pose agreement (identity and order) for every subunit
    lo_d = lib.set_build(the_map.slots, lo_s["coords"], lo_s["anc_octave"], lo_s["subv"], np.arange(len(lo_s["coords"])))
    agree = True
    for sub, hi_s in zip(subs, his):
        if not hi_s["pairs"]:
            continue
        hi_d = lib.set_build(sub.slots, hi_s["coords"], hi_s["anc_octave"], hi_s["subv"], np.arange(len(hi_s["coords"])))
        top, idx, st = lib.match_topk(hi_d, lo_d, cc, dist, k)
        agree = agree and bool(st["n_pairs"] == hi_s["pairs"] and np.array_equal(idx, hi_s["order"]) and
                               np.array_equal(top[:, 1], hi_s["res"][hi_s["order"]][:, 1]))
        hi_d.close()
    lo_d.close()
    what = "EVERY anchor of both octaves (the full step)" if whole else "the base-octave anchors"
    out = dict(value=n_corr / dt, unit="correlations/s", cores=threads, kind="port",
               sample="CPU oracle (scalar C, %d thread%s) on %s of the same workload: %d map anchors (%d rows) x %d subunits of "
                      "%s anchors (%d rows), %d pairs over cc, %.1f s" % (threads, "" if threads == 1 else "s", what, len(lo_s["coords"]), len(lo_s["dsc"]),
                                                                          len(subs), "/".join(str(len(h["coords"])) for h in his),
                                                                          sum(len(h["dsc"]) for h in his), n_pairs, dt),
               reference_python_corr_per_s=REFERENCE_PY_CORR_S,
               reference_python_note="the reference itself (python + numpy, one thread) as measured in the BUILD container on a 272 x 549-row case "
                                     "(SURVEY.md section 6 [probe]); it cannot travel to the GPU box, the C oracle is its port")
    return out, agree


# --------------------------------------------------------------------------------------------
# launching
# --------------------------------------------------------------------------------------------

def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU with torch.distributed.run as a CHILD process --
    this parent has not touched a GPU (no HIP call, no torch import) and never does; it relays the children's output and exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=sorted(WORKLOADS) + ["auto"], help="auto: c3 on one GPU, c4 (strong scaling) on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=0, choices=(0, 2, 3, 4), help="steps in flight (groups of device sets); 0 = 4 on one GPU, 3 on several, or 4 when this "
                    "rank holds a single subunit (its step is then a string of short launches, and the host's turn-around shows)")
    ap.add_argument("--serial", action="store_true", help="lanes serialised for the whole run: the mode the rocprofv3 summaries in profiles/ are "
                    "taken in, so that a kernel's average duration there is the one the roofline pass measures")
    ap.add_argument("--batched", action="store_true", help="one launch per stage for all structures of a step (mad_set_build_many) and one GEMM "
                    "grid for all its matches (mad_set_batching): less device time per step, longer dependency chains")
    ap.add_argument("--repeat-anchors", action="store_true", help="feed every step the SAME anchor lists (rounds 1-3: the library then skips the "
                    "upload work of a set rebuilt with unchanged anchors).  Default: two listings of the same anchors alternate, every step uploads")
    ap.add_argument("--rehearse-resize", type=int, default=0, metavar="ROWS", help="REHEARSAL (N > 1): after the set-up the wire images of one group's "
                    "map build are cut to ROWS rows, so that the next import overflows (MAD_ENOSPC) and the ranks have to agree on sizing them again; the "
                    "line reports map_image_resizes")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="N", help="REHEARSAL on one GPU: do the per-step work of rank 0 of an N-rank "
                    "job (its share of the map build, the import of all N shares, its subunits); the other ranks' map rows are built once, untimed.  "
                    "The line is labelled as an estimate and is not a multi-GPU measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # MAD_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (all ranks then share
    # the visible devices); the real runs use RCCL ("nccl"), one GPU per rank
    backend = os.environ.get("MAD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("MAD_DIST_FORCE", "0") == "1"      # MAD_DIST_FORCE=1: the collectives of the N > 1 path at world 1 (RCCL on one GPU)
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        patience = datetime.timedelta(seconds=int(os.environ.get("MAD_DIST_TIMEOUT_S", "300")))      # a rank that died must not hold the others for the default 10 minutes
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=patience)
        else:
            dist.init_process_group(backend, timeout=patience)

    from mad_amd import _lib
    from mad_amd import dist as mdist
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    lib = _lib.Lib(local)
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)

    # device peaks measured in this run (untimed): a streaming copy and the int8 MFMA issue rate
    peak_copy_gbs, peak_i8_tops = lib.probe_peaks() if rank == 0 else (None, None)

    emu = args.emulate_rank_of if (world == 1 and args.emulate_rank_of > 1) else 0
    wl = args.workload if args.workload != "auto" else ("c4" if (world > 1 or emu) else "c3")
    W = WORKLOADS[wl]
    cc, dist_thr, k = W["cc_threshold"], W["anchor_dist"], W["n_samples"] * W["copies"]      # k = n_samples x n_copies (MaD.py:502)
    # Which pair grids this rank works on: whole subunits round-robin, and -- when the subunits do not divide by the ranks (C5: 12 on
    # 8) -- blocks of map rows of the leftover ones inside groups of ranks (mad_amd/dist.py: plan_partition, PartitionedMatch)
    n_ranks_part = emu if emu else world
    pm = None
    if n_ranks_part > 1 and W["n_sub"] % n_ranks_part != 0 and os.environ.get("MAD_NO_PARTITION", "0") != "1":
        # (a rehearsed rank has nobody to exchange with: "local" = the asynchronous path of a node without its collectives)
        pm = mdist.PartitionedMatch(W["n_sub"], 0 if emu else rank, n_ranks_part, make_group=None if emu else dist.new_group,
                                    stand_ins="local" if emu else None)
        PM["pm"] = pm
    the_map, subs, t_setup = build_inputs(lib, W, 0 if emu else rank, emu if emu else world, items=pm.items if pm else None)
    if world > 1:      # one anchor list for everybody: the shares of the map build are indices into it
        for arr in (the_map.coords, the_map.octave, the_map.subv, the_map.index):
            t = torch.from_numpy(arr).to("cuda" if backend == "nccl" else "cpu")
            n_all = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
            dist.broadcast(n_all, 0)
            if int(n_all.item()) != t.numel():
                sys.exit("bench.py: rank %d detected %d map anchor values, rank 0 %d" % (rank, t.numel(), int(n_all.item())))
            dist.broadcast(t, 0)
            arr[...] = t.cpu().numpy()

    def barrier():
        torch.cuda.synchronize()
        lib.synchronize()
        if world > 1:
            dist.barrier()

    def exchange(tops):
        """The step's second exchange: every rank receives every subunit's top-k poses (one fused all-gather of
        n_sub x k x 23 float64, mad_amd/dist.py).  It is started asynchronously and collected one step later, so it
        overlaps the next step's kernels.  This rank's overflow word (OVERFLOW) travels with it."""
        return mdist.TopkExchange(tops, k, W["n_sub"], rank, world, status=OVERFLOW["seen"])

    def finish_exchange(x):
        got = x.finish()
        if any(x.status):      # the same words on every rank, so every rank stops at this step
            raise ResizeMapImages()
        return got

    sharded = world > 1 or emu or use_dist      # use_dist at world 1 (MAD_DIST_FORCE): export -> RCCL all-gather -> import of the one share

    def map_set():
        if sharded:
            return mdist.ShardedSetBuild(lib, the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index, rank, world,
                                         emulate=(emu, 0) if emu else None, force=True)
        return _lib.DeviceSet(lib)

    # A rank with a single subunit (C4 on 8 GPUs) runs ~35 short launches per step: there the batched launches (one per stage for
    # its share of the map + its subunit) and a fourth step in flight pay -- rehearsed: 0.345 -> 0.317 ms per step -- while a rank
    # with several subunits is better off with one launch per structure and three steps.
    # (decided from the job's shape, not from this rank's share: every rank must issue the same sequence of collectives, and the
    # depth of the pipeline is part of that sequence)
    n_ranks_job = emu if emu else world
    light = n_ranks_job > 1 and -(-W["n_sub"] // n_ranks_job) <= 1
    # (round 3: on ONE GPU a fourth step in flight pays as well since the kernels got shorter -- C3 0.825 -> 0.805 ms per step, C4 2.04
    # -> 1.98, C5 unchanged; the rehearsed rank of four on C4 is better off with three: 0.671 against 0.694)
    if args.in_flight == 0:
        args.in_flight = 4 if (light or n_ranks_job == 1) else 3
    if args.batched or (light and not BATCHED["asked"]):
        BATCHED["on"] = True
    set_groups = [[map_set()] + [_lib.DeviceSet(lib) for _ in subs] for _ in range(args.in_flight)]
    sets = set_groups[0]
    # fresh anchor lists: one structure per set and per lane (the default on one GPU); the sharded map build and the batched launches
    # keep their prepared jobs
    FRESH["on"] = not args.repeat_anchors and not sharded and not BATCHED["on"]
    if args.serial:
        lib.set_overlap(False)
    if BATCHED["on"]:
        lib.set_batching(True)
    # Setup, not a step: all groups of device sets are created, sized and given their launch-size hints here (a set sizes
    # its describe grid and a match its pair capacity from what the previous use of the same objects needed; the wire images
    # of a sharded map build are sized by a blocking first build), the way an allocator is warmed before a run.  The W
    # warm-up steps and the K timed steps that follow all do the full work.
    OVERFLOW["watch"] = world > 1
    pending = [None]

    def after_step(tops):      # the exchange of step i completes while step i + 1 runs
        if pending[0] is not None:
            last, pending[0] = pending[0], None
            finish_exchange(last)
        pending[0] = exchange(tops)

    def resize_map_images():
        """Every rank, at the same point: nothing in flight, then the blocking sizing of every group's map images (a MAX
        all-reduce each, dist.ShardedSetBuild.resize)."""
        if pending[0] is not None:
            pending[0].finish()
            pending[0] = None
        barrier()
        for grp in set_groups:
            grp[0].resize()
        OVERFLOW["seen"] = 0
        OVERFLOW["resizes"] += 1

    for attempt in range(4):
        try:
            for grp in set_groups:
                for _ in range(2):
                    finish_exchange(exchange(hot_path_step(lib, the_map, subs, cc, dist_thr, k, grp)[1]))
            if args.rehearse_resize and attempt == 0 and world > 1:      # see the option's help
                barrier()
                set_groups[1 % len(set_groups)][0].shrink_for_rehearsal(args.rehearse_resize)
            run_steps(lib, the_map, subs, cc, dist_thr, k, set_groups, args.warmup, after_step=lambda tops: finish_exchange(exchange(tops)))
            HOST_T.clear()
            barrier()
            allocs0 = lib.device_allocations()
            t0 = time.perf_counter()
            corr, tops, stats = run_steps(lib, the_map, subs, cc, dist_thr, k, set_groups, args.steps, after_step=after_step)
            last, pending[0] = pending[0], None
            gathered = finish_exchange(last) if last is not None else []      # every step's exchange completes inside the timed region
            barrier()
            dt = time.perf_counter() - t0
            allocs_timed = lib.device_allocations() - allocs0
            break
        except ResizeMapImages:
            resize_map_images()
    else:
        sys.exit("bench.py: the map's wire images overflowed again after %d re-sizings" % OVERFLOW["resizes"])
    host_overlapped = {k_: 1e3 * v / max(args.steps, 1) for k_, v in HOST_T.items()}
    if sharded:      # the pieces of build_enqueue that belong to the sharded map build, per call, over the whole run so far
        for grp in set_groups:
            for k_, v in grp[0].host_s.items():
                if k_ != "calls" and grp[0].host_s["calls"]:
                    host_overlapped["map_" + k_] = host_overlapped.get("map_" + k_, 0.0) + 1e3 * v / grp[0].host_s["calls"] / len(set_groups)

    # The same steps with the anchor lists REPEATED (what rounds 1-3 timed): how much of a step is the upload of fresh lists.
    ms_repeat = None
    if FRESH["on"]:
        FRESH["on"] = False
        n_rep = max(args.steps // 2, 4)
        run_steps(lib, the_map, subs, cc, dist_thr, k, set_groups, len(set_groups) + 1)      # every group once with the list it keeps
        barrier()
        t1 = time.perf_counter()
        run_steps(lib, the_map, subs, cc, dist_thr, k, set_groups, n_rep)
        barrier()
        ms_repeat = 1e3 * (time.perf_counter() - t1) / n_rep
        FRESH["on"] = True

    # One step at a time, lanes overlapped, nothing else in flight: what a caller who docks ONE batch waits for.
    lat = []
    for _ in range(min(args.steps, 10)):
        barrier()
        t1 = time.perf_counter()
        _, tops_l, _ = hot_path_step(lib, the_map, subs, cc, dist_thr, k, sets)
        exchange(tops_l).finish()
        lib.synchronize()
        lat.append(time.perf_counter() - t1)
    lat_all = torch.tensor([float(np.median(lat)) if lat else 0.0], dtype=torch.float64)

    # Per-kernel durations for the rooflines: in the timed region the builds of the sets and the matches
    # overlap on the device (one lane per structure), so a launch's elapsed time there depends on what ran beside it.  The same
    # steps are therefore repeated with the lanes serialised onto one stream, HIP events around every launch.
    n_serial = min(args.steps, 5)
    lib.set_overlap(False)
    lib.timing_enable(True)
    lib.timing_reset()
    HOST_T.clear()
    t1 = time.perf_counter()
    for _ in range(n_serial):
        hot_path_step(lib, the_map, subs, cc, dist_thr, k, sets)
    lib.synchronize()
    dt_serial = time.perf_counter() - t1
    lib.timing_enable(False)
    lib.set_overlap(not args.serial)

    # how much of the pair list the pruned pose search had to search exactly, per subunit (untimed; one match at a time)
    pose_sel = []
    if rank == 0 and not sharded:
        lo_x, his_x = enqueue_builds(lib, the_map, subs, sets)
        for hi_x in his_x:
            _, _, st_x = lib.match_topk(hi_x, lo_x, cc, dist_thr, k)
            pose_sel.append((int(lib.last_pose_selected()), int(st_x["n_pairs"])))

    red_dev = "cuda" if (world == 1 or backend == "nccl") else "cpu"
    t_all = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    c_all = torch.tensor([float(corr)], dtype=torch.float64, device=red_dev)
    lat_all = lat_all.to(red_dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(c_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(lat_all, op=dist.ReduceOp.MAX)
    t_max, corr_total = float(t_all.item()), float(c_all.item())
    corr_by_rank = [float(corr)]
    if world > 1:
        each = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(each, torch.tensor([float(corr)], dtype=torch.float64, device=red_dev))
        corr_by_rank = [float(e.item()) for e in each]

    # N > 1: (a) the map set assembled from the ranks' shares must be the set one GPU builds from the whole anchor list, bit
    # for bit; (b) rank 0 then runs the WHOLE workload alone (all subunits, unsharded map build), outside the timed region:
    # the one-GPU point of the strong-scaling curve measured in the same run, and the top-k every rank's result is held to.
    shard_check, one_gpu = None, None
    if sharded:
        lo_sh = sets[0].enqueue()      # a collective: every rank takes part
        lib.synchronize()
    if sharded and rank == 0:
        a = lo_sh.download()
        ref_set = lib.set_build(the_map.slots, the_map.coords, the_map.octave, the_map.subv, the_map.index)
        b = ref_set.download()
        shard_check = bool(all(np.array_equal(a[f], b[f]) for f in ("anchor", "main", "sec", "R", "dsc")))
        ref_set.close()
        _, all_subs, _ = build_inputs(lib, W, 0, 1)
        PM["pm"] = None      # (the whole workload on this one GPU: nothing is partitioned)
        was_batched = BATCHED["on"]      # the whole workload on one GPU in ITS best form: one launch per structure, three steps in flight
        BATCHED["on"] = False
        lib.set_batching(False)
        groups1 = [[_lib.DeviceSet(lib) for _ in range(1 + len(all_subs))] for _ in range(3)]
        for grp in groups1:
            for _ in range(2):
                hot_path_step(lib, the_map, all_subs, cc, dist_thr, k, grp)
        run_steps(lib, the_map, all_subs, cc, dist_thr, k, groups1, args.warmup)
        lib.synchronize()
        t1 = time.perf_counter()
        corr1, tops1, _ = run_steps(lib, the_map, all_subs, cc, dist_thr, k, groups1, args.steps)
        lib.synchronize()
        dt1 = time.perf_counter() - t1
        same = None
        if not emu:
            same = bool(len(gathered) == len(tops1) and all(np.array_equal(g, t) for g, t in zip(gathered, tops1)))
        else:      # (a rehearsed rank's blocks of leftover subunits have nobody to merge with: only its whole subunits are compared)
            same = bool(all(np.array_equal(tops1[s.item], t) for s, t in list(zip(subs, tops))[:pm.n_whole if pm else len(subs)]))
        one_gpu = dict(value=corr1 * args.steps / dt1, unit="correlations/s", ms_per_step=1e3 * dt1 / args.steps,
                       topk_identical_to_sharded_run=same,
                       note="rank 0 running the whole workload alone after the timed region (same build, same box): the N = 1 point of this strong-scaling run")
        for grp in groups1:
            for s_ in grp:
                s_.close()
        BATCHED["on"] = was_batched
        lib.set_batching(was_batched)
    if world > 1:
        dist.barrier()

    if rank == 0:
        groups = {}
        for gname in ("orient", "describe", "correlate", "pairs", "pose", "topk"):
            ms, n = lib.timing_get(gname)
            groups[gname] = dict(ms_total=ms, launches=n)
        n_anchor_lo = len(the_map.coords) if not sharded else len(mdist.share_of(len(the_map.coords), 0, emu if emu else world))
        rows_lo = stats[0]["n_lo"] if stats else 0
        rows_lo_built = rows_lo if not sharded else sets[0].share.size()[0]
        rows_hi = sum(s["n_hi"] for s in stats)
        anchors_hi = sum(len(s.coords) for s in subs)
        pairs = sum(s["n_pairs"] for s in stats)
        # one roofline entry per kernel group; counters (HBM traffic, VALU issue share) come from the committed PMC summary of THIS
        # build, identified by the fingerprint of the kernel sources -- a summary of another build is named but not used
        traffic, valu_util, counters_from = {}, {}, None
        import glob
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_c3_serial_summary.json")))      # newest round last
        pj = {}
        if wl == "c3" and world == 1 and profs:      # PMC passes of this same command (profiles/README.md)
            with open(profs[-1]) as fh:
                pj = json.load(fh)
            meta = pj.get("_meta", {})
            fresh = meta.get("source_fingerprint") == source_fingerprint()
            counters_from = dict(file=os.path.relpath(profs[-1], ROOT), git=meta.get("git"), source_fingerprint=meta.get("source_fingerprint"),
                                 this_build=source_fingerprint(), stale=not fresh)
            if not fresh:
                pj = {}
        def summary_of(prefix):      # template arguments are part of a kernel's name: the instance that took the most time
            keys = [k_ for k_ in pj if k_ != "_meta" and (k_ == prefix or k_.startswith(prefix + "<") or (prefix == "k_corr_gemm" and k_.startswith(prefix)))]
            return max(keys, key=lambda k_: pj[k_].get("total_us", 0.0)) if keys else None

        for kn, gname in (("k_pose_bounds", "pose"), ("k_describe", "describe"), ("k_orient", "orient"), ("k_corr_gemm", "correlate"),
                          ("k_pair_emit2", "pairs")):
            kn = summary_of(kn)
            if kn in pj and "fetch_size_bytes_avg" in pj[kn]:
                traffic[gname] = pj[kn]["fetch_size_bytes_avg"] + pj[kn].get("write_size_bytes_avg", 0.0)
            if kn in pj and "SQ_INSTS_VALU_avg" in pj[kn] and pj[kn].get("avg_us"):
                # share of the SIMDs' issue cycles taken by vector ALU instructions (4 cycles each on a 16-lane SIMD, 1024 SIMDs, 2.4 GHz)
                valu_util[gname] = 4.0 * pj[kn]["SQ_INSTS_VALU_avg"] / (pj[kn]["avg_us"] * 1e-6 * 2.4e9 * 1024)
        per_step = {g: max(groups[g]["launches"], 1) / n_serial for g in groups}      # launches per step
        alg = {
            "describe": ("k_describe", "hbm", DESCRIBE_BYTES * (rows_lo_built + rows_hi), HBM_PEAK_GBS, "GB/s", 1e9),
            "orient": ("k_orient", "hbm", ORIENT_BYTES * (n_anchor_lo + anchors_hi), HBM_PEAK_GBS, "GB/s", 1e9),
            "correlate": ("k_corr_gemm2", "mfma", 2.0 * 1024 * corr, I8_PEAK_TOPS, "TOP/s", 1e12),
            # pose scoring reads a pair (8 B) and writes a count (4 B); its real limit is float64 VALU + LDS latency
            "pose": ("k_pose_bounds + k_pose_lds", "hbm", 12.0 * pairs, HBM_PEAK_GBS, "GB/s", 1e9),
            "pairs": ("k_pair_count + k_pair_emit2", "hbm", 8.0 * corr + 16.0 * pairs, HBM_PEAK_GBS, "GB/s", 1e9),
            "topk": ("top-k kernels", "hbm", 12.0 * pairs, HBM_PEAK_GBS, "GB/s", 1e9),
        }
        roofs = {}
        for gname, (kname, bound, work_per_step, peak, unit, scale) in alg.items():
            ms = groups[gname]["ms_total"] / max(groups[gname]["launches"], 1)
            work = work_per_step / per_step[gname]
            roofs[gname] = dict(kernel=kname, bound=bound, achieved=work / (ms * 1e-3) / scale if ms > 0 else 0.0, peak=peak, unit=unit,
                                traffic=traffic.get(gname), avg_launch_ms=ms, ms_per_step=groups[gname]["ms_total"] / n_serial)
            roofs[gname]["frac"] = roofs[gname]["achieved"] / peak
            # against the peak MEASURED in this run on this device: a streaming copy for the HBM-bound kernels, the bare int8 MFMA loop for the GEMM
            pk = peak_copy_gbs if bound == "hbm" else peak_i8_tops
            roofs[gname]["peak_measured"] = pk
            roofs[gname]["frac_of_measured"] = roofs[gname]["achieved"] / pk if pk else None
            if gname in valu_util:
                roofs[gname]["valu_issue_share"] = round(valu_util[gname], 3)      # from the SQ counters of the profiled run (profiles/)
        # `roofline` = the HBM stage SURVEY.md 8(d) names as the binding roofline of the headline metric (the texel
        # gathers of orient + describe), represented by its larger kernel, k_describe.  The pose search takes a comparable
        # share of the device time but is bound by VALU issue and LDS latency, which an hbm | mfma roofline
        # cannot express: it is listed under `others` with its own note.
        dom_name = "describe"
        roof = dict(roofs[dom_name])
        roof["counters_from"] = counters_from
        roof["kernel_ms_per_step"] = {g: groups[g]["ms_total"] / n_serial for g in groups}
        roof["timing"] = ("HIP events around every launch over %d steps with the lanes serialised onto one stream (%.3f ms/step); "
                          "the timed region overlaps the lanes (%.3f ms/step)" % (n_serial, 1e3 * dt_serial / n_serial, 1e3 * t_max / args.steps))
        roof["host_ms_per_step"] = {k_: 1e3 * v / n_serial for k_, v in HOST_T.items()}
        roof["host_ms_per_step_timed_region"] = host_overlapped
        t_hbm = (groups["orient"]["ms_total"] + groups["describe"]["ms_total"]) / n_serial * 1e-3
        b_hbm = ORIENT_BYTES * (n_anchor_lo + anchors_hi) + DESCRIBE_BYTES * (rows_lo_built + rows_hi)
        roof["hbm_stage"] = dict(kernels="k_orient + k_describe", algorithmic_bytes_per_step=b_hbm, seconds_per_step=t_hbm,
                                 achieved=b_hbm / t_hbm / 1e9 if t_hbm > 0 else 0.0, unit="GB/s", frac=(b_hbm / t_hbm / 1e9) / HBM_PEAK_GBS if t_hbm > 0 else 0.0)
        roof["hbm_stage"]["peak_measured"] = peak_copy_gbs
        roof["hbm_stage"]["frac_of_measured"] = roof["hbm_stage"]["achieved"] / peak_copy_gbs if peak_copy_gbs else None
        roof["peaks"] = dict(hbm_spec_GBs=HBM_PEAK_GBS, hbm_copy_measured_GBs=peak_copy_gbs, i8_mfma_spec_TOPs=I8_PEAK_TOPS, i8_mfma_measured_TOPs=peak_i8_tops,
                             how="mad_probe_peaks, untimed, same process and device: a 1 GiB -> 1 GiB streaming copy kernel (read + write bytes) and "
                                 "v_mfma_i32_16x16x64_i8 from registers, 8 independent accumulators per wave, 4 waves per SIMD; spec figures from MI355X_MICROARCH.md")
        roof["largest_share_of_device_time"] = max(groups, key=lambda g: groups[g]["ms_total"])
        if roof.get("traffic") and roof.get("avg_launch_ms"):      # what the kernel really moved (FETCH_SIZE + WRITE_SIZE of the profiled run) over the same duration
            roof["achieved_from_counters"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
            roof["frac_from_counters"] = roof["achieved_from_counters"] / HBM_PEAK_GBS
        roof["note"] = ("achieved = SURVEY 8(d)'s ALGORITHMIC 51 200 B per row (4 096 samples x 12 B gathered + 2 048 B written) over the launch time. "
                        "Since round 3 the kernel gathers 4-byte texels (a quantised unit direction; 16 KB per row) and fetches the 16-byte texel "
                        "only for the 3-4 % of samples its table classifier leaves open, so the bytes it really moves are about a third of the "
                        "algorithmic figure; anchors are worked on in Morton order and neighbouring rows meet in the XCD's L2.  What it waits for is "
                        "not HBM but the per-lane request path of fully divergent loads and the chain of phases of a row (DESIGN.md section 6c)")
        l_hi_mean = float(np.mean([s["l_hi"] for s in stats])) if stats else 0.0
        pts = pairs * l_hi_mean      # transformed hi-cloud points per step
        t_pose = groups["pose"]["ms_total"] / n_serial * 1e-3
        roofs["pose"]["note"] = ("not an HBM kernel (the HBM figure is its algorithmic 12 B/pair): every transformed hi point (%.3g per step, %.0f G/s) "
                                 "goes through a float32 occupancy-bitmap test; vector-ALU issue is the binding resource (valu_issue_share, "
                                 "SQ_INSTS_VALU of the profiled run)" % (pts, pts / t_pose / 1e9 if t_pose > 0 else 0.0))
        roof["others"] = {g: roofs[g] for g in roofs if g != dom_name}

        cpu, agree, cpu_all, agree_whole = (None, None, None, None)
        if world == 1 and not emu and not args.no_cpu_baseline:
            cpu, agree = cpu_baseline(the_map, subs[:4], cc, dist_thr, k, lib)
            n_host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            n_host = min(n_host, 16)      # a one-GPU box's CPU share
            if n_host > 1:      # the FULL step (every anchor, both octaves) on the host cores of this box's share: also the top-k check of the line
                cpu_all, agree_whole = cpu_baseline(the_map, subs[:4], cc, dist_thr, k, lib, threads=n_host, whole=True)
        refine_line = None
        if world == 1 and not emu:
            lib.timing_enable(True)
            refine_line = refine_ccc_leg(lib, the_map, subs[:4], tops[:4], W)
            lib.timing_enable(False)
            roof["others"].update(refine_line.pop("roofline"))

        # The workload rounds 1-2 timed (C3 WITHOUT its noise: more than twice the pairs per match), a few steps of it beside the headline so
        # that this round's line can be held against theirs; not part of `value`.
        side_clean = None
        if world == 1 and not emu and wl == "c3" and not args.no_cpu_baseline and "c3clean" in WORKLOADS:
            Wc = WORKLOADS["c3clean"]
            map_c, subs_c, _ = build_inputs(lib, Wc)
            groups_c = [[_lib.DeviceSet(lib) for _ in range(1 + len(subs_c))] for _ in range(args.in_flight)]
            kc = Wc["n_samples"] * Wc["copies"]
            for grp in groups_c:
                for _ in range(2):
                    hot_path_step(lib, map_c, subs_c, Wc["cc_threshold"], Wc["anchor_dist"], kc, grp)
            n_c = max(args.steps, 4)      # (as many as the headline: the pipeline's fill and drain weigh the same)
            run_steps(lib, map_c, subs_c, Wc["cc_threshold"], Wc["anchor_dist"], kc, groups_c, len(groups_c) + args.warmup)
            passes, allocs_c = [], []      # two passes, the faster one reported (one run of eight showed a 60 ms stall in the single pass there was then; no device allocation in either pass since)
            for _ in range(2):
                barrier()
                a1 = lib.device_allocations()
                t1 = time.perf_counter()
                corr_c, _, _ = run_steps(lib, map_c, subs_c, Wc["cc_threshold"], Wc["anchor_dist"], kc, groups_c, n_c)
                barrier()
                passes.append(time.perf_counter() - t1)
                allocs_c.append(lib.device_allocations() - a1)
            dt_c = min(passes)
            side_clean = dict(workload="c3clean: C3 without its Gaussian noise, what BENCH_r01 / r02 timed", steps=n_c, ms_per_step=1e3 * dt_c / n_c,
                              ms_per_step_each_pass=[1e3 * x / n_c for x in passes], device_allocations_each_pass=allocs_c, value=corr_c * n_c / dt_c, unit="correlations/s")
            for st_ in [map_c] + subs_c:
                st_.ms.release_device()

        n_ranks = emu if emu else world
        line = {
            "metric": "anchor-pair x rotation correlations/sec on 256^3 map; top-k pose agreement",
            "value": corr_total * args.steps / t_max,
            "unit": "correlations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "latency_ms_single_step": 1e3 * float(lat_all.item()),
            "ms_per_step_repeated_anchor_lists": ms_repeat,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "i8 (correlation, exact int32 accumulate) / f64 (binning, pose scoring)",
            "data": "synthetic",
            "config": {"workload": W["desc"], "workload_file": W["file"], "noise_sigma_of_max": W.get("noise", 0.0), "copies_per_subunit": W["copies"], "map": "%d^3 @ %.1f A/voxel, %.0f A" % (W["N"], W["vs"], W["res"]),
                       "subunits": W["n_sub"], "subunits_this_rank": len(subs), "map_anchors": len(the_map.coords), "map_rows": rows_lo,
                       "map_anchors_built_by_this_rank": n_anchor_lo, "map_rows_built_by_this_rank": rows_lo_built,
                       "subunit_anchors_this_rank": anchors_hi, "subunit_rows_this_rank": rows_hi, "pairs_over_cc_this_rank": pairs, "pairs_over_cc_per_correlation": pairs / corr if corr else None,
                       "pose_selected_frac": (sum(a for a, _ in pose_sel) / max(sum(b for _, b in pose_sel), 1)) if pose_sel else None,
                       "pose_selected_per_subunit": pose_sel or None,
                       "cc_threshold": cc, "top_k": k, "correlations_per_step": corr_total, "correlations_per_step_this_rank": corr,
                       "parallelism": ("1 process per GPU over RCCL: map anchors dealt round-robin (orient + describe), all-gather of the rows; "
                                       "subunits dealt round-robin (correlate + pose + top-k), all-gather of the top-k poses") if world > 1 else "single GPU",
                       "partition": None if pm is None else dict(units_of_this_rank=[list(u) for u in pm.mine], groups=pm.groups, load_in_subunits=pm.load(),
                                                                  note="whole subunits round-robin; each leftover subunit's pair grid split by blocks of map rows "
                                                                       "inside a group of ranks (sharded_match: OR all-reduce of the cloud flags, all-gather + merge of the per-block top-k)"),
                       "correlations_per_step_by_rank": corr_by_rank,
                       "max_over_mean_rank_correlations": (max(corr_by_rank) / (sum(corr_by_rank) / len(corr_by_rank))) if sum(corr_by_rank) > 0 else None,
                       "anchor_lists": ("fresh every step: two listings of each structure's anchors alternate, so every mad_set_build uploads, "
                                        "canonicalises duplicates and Morton-sorts its list inside the timed region (host_ms_per_step_timed_region.build_enqueue); "
                                        "ms_per_step_repeated_anchor_lists = the same steps with the lists repeated, as rounds 1-3 timed them") if FRESH["on"]
                                       else "repeated: every step feeds the same lists, the library skips the upload of unchanged anchors",
                       "pipelining": "%d steps in flight: the builds (and, with 3, the matches) of the next step are enqueued before the results of a step are awaited; every step does the full work" % args.in_flight,
                       "launches": ("batched: one launch per stage for all structures of a step, one GEMM grid for all its matches" if BATCHED["on"]
                                    else "one mad_set_build per structure and one GEMM per match, each on its own lane (batched alternative: --batched)"),
                       "latency_note": "latency_ms_single_step = one step submitted alone (lanes overlapped, nothing else in flight, result exchange included), median of %d; ms_per_step = throughput of identical steps, %d in flight" % (len(lat), args.in_flight),
                       "topk_agrees_with_cpu_oracle": agree_whole if agree_whole is not None else agree,
                       "topk_agrees_with_cpu_oracle_on": None if agree is None else ("every anchor of both octaves, all subunits (the full step) and the base-octave sample" if agree_whole is not None else "the base-octave sample"),
                       "topk_agrees_on_sample": agree,
                       "sharded_map_set_identical_to_unsharded": shard_check,
                       "device_allocations_in_timed_region": allocs_timed,      # buffers the library (re)allocated inside the timed steps: a steady state has none
                       "map_image_resizes": OVERFLOW["resizes"],      # times the ranks agreed to size the map's wire images again (0 unless rehearsed)
                       "setup_s": t_setup,
                       "setup_detail_s": {k_: round(v, 4) for k_, v in SETUP_T.items()}},
            "roofline": roof,
            # cpu_baseline = the oracle on the SAME work as the timed step (every anchor of both octaves, all subunits) on the host cores of
            # this box's share; the one-core figure is a bounded sample (the base-octave anchors: the full step takes minutes on one core)
            "cpu_baseline": cpu_all if cpu_all is not None else cpu,
            "cpu_baseline_one_core_sample": cpu if cpu_all is not None else None,
            "refine_ccc": refine_line,
            "c3_without_noise": side_clean,
            "one_gpu_same_workload": one_gpu,
        }
        if emu:
            line["rehearsal"] = ("ESTIMATE, not a multi-GPU measurement: one GPU doing the per-step work of rank 0 of %d (its share of the map build, the "
                                 "import of all %d shares, %d of %d subunits); value = that rank's correlations x %d / its time, without any link traffic"
                                 % (n_ranks, n_ranks, len(subs), W["n_sub"], n_ranks))
            line["value"] = corr * n_ranks * args.steps / t_max
        if one_gpu is not None:
            line["speedup_vs_one_gpu_same_workload"] = line["value"] / one_gpu["value"]
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    lib.close()


if __name__ == "__main__":
    main()
