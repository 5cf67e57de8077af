"""GPU parity tests, stage by stage: the HIP path (through the C-ABI) against the CPU
oracle on the same seeded inputs.  Integer outputs must be identical; float64 outputs
agree to the stated tolerance."""
import numpy as np
import pytest

from mad_amd import synth
from mad_amd.eqsp import EQSP_Sphere
from oracle import oracle as O

pytestmark = pytest.mark.gpu

E112 = EQSP_Sphere(112)
E16 = EQSP_Sphere(16)


def _field(shape, seed, hollow=0.25):
    vol = synth.blob_volume(shape, n_blobs=40, seed=seed, sigma=(1.5, 3.5), hollow=hollow)
    g = synth.gradient_field(vol)
    return vol, np.ascontiguousarray(g[..., 0]), np.ascontiguousarray(g[..., 1]), np.ascontiguousarray(g[..., 2])


@pytest.fixture(scope="module")
def fields(lib):
    out = {}
    for octave, shape, seed in ((1, (40, 44, 48), 1), (0, (60, 64, 70), 2)):
        vol, gx, gy, gz = _field(shape, seed)
        slot = lib.new_slot()
        lib.upload_field(slot, np.stack([gx, gy, gz]))
        out[octave] = dict(slot=slot, gx=gx, gy=gy, gz=gz, shape=shape)
    return out


def _anchors(shape, octave, n, seed):
    # include border cases that the reference rejects (Orientator.py:131-135)
    margin = 8 if octave == 1 else 16
    inner = synth.interior_anchors(shape, n, margin + 1, seed)
    edge = np.array([[margin - 1, shape[1] // 2, shape[2] // 2], [shape[0] // 2, shape[1] - margin - 1, shape[2] // 2],
                     [margin, margin, margin], [shape[0] - margin - 2, shape[1] - margin - 2, shape[2] - margin - 2]], np.int32)
    return np.concatenate([inner[: n // 2], edge, inner[n // 2:]]).astype(np.int32)


@pytest.mark.parametrize("octave", [1, 0])
def test_orient_matches_oracle(lib, fields, octave):
    f = fields[octave]
    coords = _anchors(f["shape"], octave, 60, 10 + octave)
    ref = O.orient(f["gx"], f["gy"], f["gz"], octave, coords, E112.sphere_eqsp, E112.p_centers_eqsp)
    got = lib.orient(f["slot"], octave, coords)
    assert len(ref["anchor"]) > 20, "test input produces too few rows to mean anything"
    assert got["n_reject"] == ref["n_reject"] and ref["n_reject"] >= 2
    np.testing.assert_array_equal(got["anchor"], ref["anchor"])
    np.testing.assert_array_equal(got["main"], ref["main"])
    np.testing.assert_array_equal(got["sec"], ref["sec"])
    np.testing.assert_array_equal(got["counts"], ref["counts"])
    np.testing.assert_allclose(got["R"], ref["R"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("octave", [1, 0])
def test_describe_matches_oracle(lib, fields, octave):
    f = fields[octave]
    coords = _anchors(f["shape"], octave, 40, 20 + octave)
    rows = O.orient(f["gx"], f["gy"], f["gz"], octave, coords, E112.sphere_eqsp, E112.p_centers_eqsp)
    rc = coords[rows["anchor"]]
    R = rows["R"]
    # add rows whose sample cube leaves the grid (-> all-zero descriptor, Descriptor.py:140-149) and identity rotations
    extra_c = np.array([[2, 3, 4], [f["shape"][0] - 3, 20, 20], rc[0], rc[1]], np.int32)
    extra_R = np.stack([R[0], R[1], np.identity(3), np.identity(3)])
    rc = np.concatenate([rc, extra_c])
    R = np.concatenate([R, extra_R])
    ref = O.describe(f["gx"], f["gy"], f["gz"], octave, rc, R, E16.sphere_eqsp)
    got = lib.describe(f["slot"], octave, rc, R)
    assert ref[:-4].sum() > 0 and ref[-4].sum() == 0 and ref[-3].sum() == 0 and ref[-1].sum() > 0
    np.testing.assert_array_equal(got, ref)


def _bound_hugging_field(shape, table, seed):
    """A gradient field whose every voxel points (to float32 accuracy) at a bound of the EQSP table: theta on a zone edge
    or phi on a belt edge, displaced by 0, +-1e-7 ... +-1e-4 rad, including the theta = 0 / 2 pi seam and the caps."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    b = table[rng.integers(0, len(table), n)]      # rows [theta_min, phi_min, theta_max, phi_max]
    eps = rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5, 1e-4, -1e-4], n)
    on_theta = rng.random(n) < 0.5
    lo = rng.random(n) < 0.5
    theta = np.where(on_theta, np.where(lo, b[:, 0], b[:, 2]) + eps, rng.uniform(b[:, 0], b[:, 2]))
    phi = np.where(on_theta, rng.uniform(b[:, 1], np.minimum(b[:, 3], np.pi)), np.where(lo, b[:, 1], b[:, 3]) + eps)
    # not the poles themselves: a polar direction is turned by to_dom_mat onto the meridian of the main bin's centre, which is
    # exactly a zone edge in belts with twice as many zones, and which side of it atan2 then reports is a matter of the last
    # bit of the math library (device libm, glibc and numpy differ there) -- outside what any implementation can pin
    phi = np.clip(phi, 0.02, np.pi - 0.02)
    mag = rng.uniform(0.01, 2.0, n)
    g = np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)]) * mag
    return np.ascontiguousarray(g.reshape((3,) + tuple(shape)).astype(np.float32))


def test_classifier_on_the_bounds(lib):
    """The exact classification has a cross-product tier in front of atan2 / acos (eqsp_tier2): fields made of directions
    that sit on the table's bounds must still give the oracle's histograms bit for bit -- the Orientator's float32 first
    pass and float64 rotated pass on the 112-zone table, the Descriptor on the 16-zone table."""
    shape = (44, 44, 44)
    for table, seed in ((E112.sphere_eqsp, 5), (E16.sphere_eqsp, 6)):
        g = _bound_hugging_field(shape, np.asarray(table, dtype=np.float64), seed)
        slot = lib.new_slot()
        lib.upload_field(slot, g)
        try:
            coords = synth.interior_anchors(shape, 40, 14, seed)
            ref = O.orient(g[0], g[1], g[2], 1, coords, E112.sphere_eqsp, E112.p_centers_eqsp)
            got = lib.orient(slot, 1, coords)
            for key in ("anchor", "main", "sec", "counts"):
                np.testing.assert_array_equal(got[key], ref[key])
            rng = np.random.default_rng(seed)
            R = np.stack([np.identity(3)] * 20 + [synth.random_rotation(rng) for _ in range(20)])
            want = O.describe(g[0], g[1], g[2], 1, coords, R, E16.sphere_eqsp)
            have = lib.describe(slot, 1, coords, R)
            np.testing.assert_array_equal(have, want)
            assert want.sum() > 1000
        finally:
            lib.free_field(slot)


def _random_descriptors(n, seed, base=None):
    rng = np.random.default_rng(seed)
    d = np.zeros((n, 1024), np.int16)
    for i in range(n):
        # ~27 % non-zero entries, sub-cube sums <= 64 (what descriptors look like)
        for sub in range(64):
            k = rng.integers(0, 65)
            z = rng.integers(0, 16, size=k)
            np.add.at(d[i], sub * 16 + z, 1)
    if base is not None:      # correlated copies so that some pairs pass the threshold
        m = min(n, len(base))
        noise = rng.integers(-1, 2, size=(m, 1024))
        d[:m] = np.clip(base[:m].astype(np.int64) + noise * (rng.random((m, 1024)) < 0.3), 0, 64)
    return d


def test_correlate_matches_oracle(lib):
    lo = _random_descriptors(300, 5)
    hi = _random_descriptors(140, 6, base=lo[50:])
    hi[3] = 0      # zero row: stays un-normalised, correlates to 0 (MaD.py:416)
    for cc in (0.6, 0.3):
        rh, rl, rs, _ = O.correlate(hi, lo, cc)
        gh, gl, gs = lib.correlate(hi, lo, cc)
        assert len(rh) > 50
        np.testing.assert_array_equal(gh, rh)
        np.testing.assert_array_equal(gl, rl)
        np.testing.assert_allclose(gs, rs, rtol=1e-12, atol=0)


@pytest.mark.parametrize("n_hi, n_lo", [(1, 1), (129, 513), (257, 1300), (384, 2049), (1100, 16001)])
def test_correlate_on_ragged_sizes(lib, n_hi, n_lo):
    """Row counts around the GEMM's tile edges: one row; one row past a 128-row block (a tile row that is mostly padding); a hi set
    whose last block of 128 rows is odd (its tile row is a half tile); one column past a tile; and a matrix of 625 tiles on 512
    workgroups -- whole rounds of 256 x 128 tiles, the leftover dealt as halves, every workgroup starting its next tile under the
    epilogue of the current one.  Pair list and scores are the oracle's.
    (cc is not the round 0.6: among the 17.6 M scores of the largest case one IS 0.6 -- dot 4 593 over norms whose product is 7 655 --
    and a score that equals the threshold is where dot / (|h| |l|) and the reference's sum over normalised rows may fall on either
    side of it; tools/debug_pairs.py shows the pair.)"""
    cc = 0.60000013
    lo = _random_descriptors(n_lo, 31)
    hi = _random_descriptors(n_hi, 32, base=lo[n_lo // 3:])
    rh, rl, rs, _ = O.correlate(hi, lo, cc)
    gh, gl, gs = lib.correlate(hi, lo, cc)
    assert len(rs) == 0 or np.min(np.abs(rs - cc)) > 1e-12      # no score on the threshold
    assert len(rh) >= min(n_hi, n_lo - n_lo // 3) // 2      # the planted copies
    np.testing.assert_array_equal(gh, rh)
    np.testing.assert_array_equal(gl, rl)
    np.testing.assert_allclose(gs, rs, rtol=1e-12, atol=0)


@pytest.mark.parametrize("cc", [-1.0, 0.2])
def test_correlate_rows_with_more_candidates_than_the_pair_list_holds(lib, cc):
    """The pair kernels gather a row's flagged columns into a list of 1 024 per pass of 8 192 columns (pair_list_row) and settle one
    candidate per lane; a pass with more candidates than that falls back to every lane walking its own mask words.  cc = -1: every
    entry of 64 x 9 000 is a pair, in row-major order; cc = 0.2: most are."""
    lo = _random_descriptors(9000, 41)
    hi = _random_descriptors(64, 42, base=lo[100:])
    rh, rl, rs, _ = O.correlate(hi, lo, cc)
    assert len(rh) > 64 * 2000 and (len(rs) == 0 or np.min(np.abs(rs - cc)) > 1e-12)
    gh, gl, gs = lib.correlate(hi, lo, cc)
    np.testing.assert_array_equal(gh, rh)
    np.testing.assert_array_equal(gl, rl)
    np.testing.assert_allclose(gs, rs, rtol=1e-12, atol=0)


def test_correlate_threshold_on_a_score(lib):
    """cc within 1e-12 ... 1e-6 (relative) of the score of existing pairs, on either side: the GEMM's float32 candidate test
    in front of the exact float64 comparison must not lose a pair that sits just above the threshold, nor the exact test admit
    one just below.  (At one ulp the comparison is not defined: dot / (|h| |l|) here, a dgemm over normalised rows in the
    reference, 1e-15 apart -- SURVEY.md a11.)"""
    lo = _random_descriptors(260, 15)
    hi = _random_descriptors(120, 16, base=lo[40:])
    hi[7] = lo[99]      # an identical pair: score exactly 1.0
    nh, nl = np.linalg.norm(hi.astype(np.float64), axis=1), np.linalg.norm(lo.astype(np.float64), axis=1)
    _, _, rs, _ = O.correlate(hi, lo, 0.3)
    picks = list(np.quantile(rs, [0.1, 0.5, 0.9], method="nearest")) + [1.0]
    for sc in picks:
        for cc in [sc * (1.0 + sgn * rel) for rel in (1e-12, 1e-9, 1e-7, 1e-6) for sgn in (-1.0, 1.0)]:
            rh, rl, rsc, _ = O.correlate(hi, lo, cc)
            gh, gl, gs = lib.correlate(hi, lo, cc)
            np.testing.assert_array_equal(gh, rh)
            np.testing.assert_array_equal(gl, rl)
            assert (len(rsc) == 0 or rsc.min() > cc)
    assert nh[7] == nl[99]


def test_correlate_rejects_out_of_range(lib):
    from mad_amd._lib import MadBackendError
    hi = np.full((4, 1024), 200, np.int16)
    with pytest.raises(MadBackendError, match="int8"):
        lib.correlate(hi, hi, 0.5)


def _pose_inputs(seed, n_hi_anchor=70, n_lo_anchor=160, n_pairs=4000):
    rng = np.random.default_rng(seed)
    lo_anchor_p = rng.uniform(0, 90, size=(n_lo_anchor, 3))
    # hi anchors = a rigidly moved subset of the lo anchors + jitter, so that good poses exist
    Rt = synth.random_rotation(rng)
    sub = rng.choice(n_lo_anchor, n_hi_anchor, replace=False)
    hi_anchor_p = (lo_anchor_p[sub] - 45.0) @ Rt.T + rng.normal(scale=0.8, size=(n_hi_anchor, 3))
    n_hi, n_lo = 3 * n_hi_anchor, 3 * n_lo_anchor
    hi_row_anchor = rng.integers(0, n_hi_anchor, n_hi)
    lo_row_anchor = rng.integers(0, n_lo_anchor, n_lo)
    hi_R = np.stack([synth.random_rotation(rng) for _ in range(n_hi)])
    lo_R = np.stack([synth.random_rotation(rng) for _ in range(n_lo)])
    # plant consistent rotations for a few pairs: inv(lo_R) @ hi_R == Rt.T-ish pose
    pair_hi = rng.integers(0, n_hi, n_pairs).astype(np.int32)
    pair_lo = rng.integers(0, n_lo, n_pairs).astype(np.int32)
    for t in range(0, n_pairs, 7):
        a = rng.integers(0, n_hi_anchor)
        ih = int(np.flatnonzero(hi_row_anchor == a)[0]) if np.any(hi_row_anchor == a) else 0
        il_c = np.flatnonzero(lo_row_anchor == sub[hi_row_anchor[ih]])
        if len(il_c) == 0:
            continue
        il = int(il_c[0])
        hi_R[ih] = lo_R[il] @ Rt.T
        pair_hi[t], pair_lo[t] = ih, il
    order = np.lexsort((pair_lo, pair_hi))
    pair_hi, pair_lo = pair_hi[order], pair_lo[order]
    score = rng.uniform(0.6, 1.0, n_pairs)
    hi_meta = np.stack([hi_row_anchor, rng.integers(0, 2, n_hi), rng.integers(0, 112, n_hi)], 1).astype(np.int32)
    lo_meta = np.stack([lo_row_anchor, rng.integers(0, 2, n_lo), rng.integers(0, 112, n_lo)], 1).astype(np.int32)
    hi_p, lo_p = hi_anchor_p[hi_row_anchor], lo_anchor_p[lo_row_anchor]
    hi_cloud = np.unique(hi_p[np.unique(pair_hi)], axis=0)
    lo_cloud = np.unique(lo_p[np.unique(pair_lo)], axis=0)
    return dict(pair_hi=pair_hi, pair_lo=pair_lo, pair_score=score, hi_p=hi_p, hi_R=hi_R, hi_meta=hi_meta,
                lo_p=lo_p, lo_R=lo_R, lo_meta=lo_meta, hi_cloud=hi_cloud, lo_cloud=lo_cloud)


def test_pose_score_matches_oracle(lib):
    a = _pose_inputs(3)
    ref_res, ref_cnt = O.pose_score(**a, dist=4.0)
    got_res, got_cnt = lib.pose_score(**a, dist=4.0)
    assert lib.last_pose_kernel() == 0
    assert ref_cnt.max() > 20, "planted poses should bring many anchors into coincidence"
    np.testing.assert_array_equal(got_cnt, ref_cnt)
    np.testing.assert_allclose(got_res, ref_res, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n_lo_anchor,n_pairs,box", [(5600, 600, 170.0), (9600, 300, 200.0)])
def test_pose_score_large_lo_cloud(lib, n_lo_anchor, n_pairs, box):
    """Lo clouds too large for float64 points in LDS.  5 600 points: the float32-tier kernel (k_pose_lds32) with its exact
    float64 re-test; 9 600 points: the global cell list (k_pose).  Both must give the oracle's counts bit for bit."""
    rng = np.random.default_rng(17)
    n_hi_anchor = 300
    lo_anchor_p = rng.uniform(0, box, size=(n_lo_anchor, 3))
    Rt = synth.random_rotation(rng)
    sub = rng.choice(n_lo_anchor, n_hi_anchor, replace=False)
    hi_anchor_p = (lo_anchor_p[sub] - box / 2) @ Rt.T + rng.normal(scale=1.5, size=(n_hi_anchor, 3))
    lo_R = np.stack([synth.random_rotation(rng) for _ in range(n_lo_anchor)])
    hi_R = np.stack([synth.random_rotation(rng) for _ in range(n_hi_anchor)])
    pair_hi = rng.integers(0, n_hi_anchor, n_pairs).astype(np.int32)
    pair_lo = rng.integers(0, n_lo_anchor, n_pairs).astype(np.int32)
    for t in range(0, n_pairs, 5):      # planted poses: many hi anchors land within ~dist of a lo anchor
        ih = int(rng.integers(0, n_hi_anchor))
        hi_R[ih] = lo_R[sub[ih]] @ Rt.T
        pair_hi[t], pair_lo[t] = ih, sub[ih]
    order = np.lexsort((pair_lo, pair_hi))
    pair_hi, pair_lo = pair_hi[order], pair_lo[order]
    meta_h = np.stack([np.arange(n_hi_anchor), np.ones(n_hi_anchor), np.zeros(n_hi_anchor)], 1).astype(np.int32)
    meta_l = np.stack([np.arange(n_lo_anchor), np.ones(n_lo_anchor), np.zeros(n_lo_anchor)], 1).astype(np.int32)
    a = dict(pair_hi=pair_hi, pair_lo=pair_lo, pair_score=rng.uniform(0.6, 1.0, n_pairs), hi_p=hi_anchor_p, hi_R=hi_R, hi_meta=meta_h,
             lo_p=lo_anchor_p, lo_R=lo_R, lo_meta=meta_l, hi_cloud=np.unique(hi_anchor_p, axis=0), lo_cloud=np.unique(lo_anchor_p, axis=0))
    ref_res, ref_cnt = O.pose_score(**a, dist=4.0)
    got_res, got_cnt = lib.pose_score(**a, dist=4.0)
    assert lib.last_pose_kernel() == (1 if n_lo_anchor == 5600 else 2)      # the kernel this case is meant for
    assert ref_cnt.max() > 100 and ref_cnt.mean() > 1
    np.testing.assert_array_equal(got_cnt, ref_cnt)
    np.testing.assert_allclose(got_res, ref_res, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n_lo_anchor,box", [(900, 150.0), (5600, 170.0)])
def test_pose_score_threshold_shell(lib, n_lo_anchor, box):
    """The occupancy-bitmap prefilter of the search kernels must never drop a point that the float64 test would count:
    with identity poses the hi cloud IS the set of transformed points, and it is built to sit on the decision surface --
    at distance dist -/+ {1e-12 .. 0.5} from a lo point in random directions, on voxel faces and corners of the bitmap
    lattice (multiples of 0.8 A from the cloud's bounding box), and beyond the box."""
    rng = np.random.default_rng(23)
    lo_p = np.unique(np.round(rng.uniform(0, box, size=(n_lo_anchor, 3)), 3), axis=0)
    n_lo_anchor = len(lo_p)
    pts = []
    for eps in (1e-12, 1e-9, 1e-6, 1e-4, 1e-3, 1e-2, 0.03, 0.1, 0.5):
        for sign in (-1.0, 1.0):
            c = lo_p[rng.integers(0, n_lo_anchor, 60)]
            u = rng.normal(size=(60, 3))
            u /= np.linalg.norm(u, axis=1)[:, None]
            pts.append(c + u * (4.0 + sign * eps))
    for off in ([4.0, 0, 0], [0, -4.0, 0], [0, 0, 4.0], [2.4, 3.2, 0], [0, -2.4, 3.2], [4.0, 0, 0.0000001]):      # exactly dist: sqrt(d2) < dist is false
        pts.append(lo_p[rng.integers(0, n_lo_anchor, 40)] + np.array(off))
    lattice = lo_p.min(0) + 0.8 * rng.integers(-12, int(box / 0.8) + 12, size=(400, 3))      # voxel corners, some outside
    pts.append(lattice)
    pts.append(lattice + 1e-6)
    pts.append(lattice + np.array([0.4, 0.0, 0.0]))
    pts.append(rng.uniform(-30, box + 30, size=(300, 3)))
    hi_p = np.unique(np.concatenate(pts), axis=0)
    n_hi = len(hi_p)
    eye = np.tile(np.identity(3), (8, 1, 1))
    # pairs: identity pose (hi anchor 0 onto itself: R = I, p_hi = p_lo) and pure translations by voxel fractions
    shifts = np.array([[0, 0, 0], [0.8, 0, 0], [0.4, 0.4, 0.4], [1e-7, -1e-7, 0], [3.2, -1.6, 0.8], [box, 0, 0], [0.79999, 0.8, 0.80001], [-0.4, 0.2, 0.1]])
    hi_anchor = np.concatenate([hi_p[:1] for _ in shifts])
    lo_anchor = hi_anchor + shifts
    meta = np.stack([np.arange(8), np.ones(8), np.zeros(8)], 1).astype(np.int32)
    a = dict(pair_hi=np.arange(8, dtype=np.int32), pair_lo=np.arange(8, dtype=np.int32), pair_score=np.full(8, 0.7), hi_p=hi_anchor, hi_R=eye,
             hi_meta=meta, lo_p=lo_anchor, lo_R=eye, lo_meta=meta, hi_cloud=hi_p, lo_cloud=lo_p)
    ref_res, ref_cnt = O.pose_score(**a, dist=4.0)
    got_res, got_cnt = lib.pose_score(**a, dist=4.0)
    assert ref_cnt[0] > 400 and ref_cnt[0] < n_hi - 400      # the shell is split by the threshold
    np.testing.assert_array_equal(got_cnt, ref_cnt)


def test_topk_order(lib):
    rng = np.random.default_rng(0)
    for n, k, hi in ((5000, 60, 40), (777, 777, 5), (20000, 840, 3), (50, 200, 9)):
        counts = rng.integers(0, hi, n).astype(np.int32)
        ref = O.topk(counts, k)
        got = lib.topk(counts, k)
        np.testing.assert_array_equal(got, ref)
        # the oracle's order is python's stable sort by count, descending (MaD.py:480)
        py = sorted(range(n), key=lambda i: counts[i], reverse=True)[: min(k, n)]
        np.testing.assert_array_equal(ref, py)


def test_topk_large_lists(lib):
    """> 2^20 pairs, many chunks per workgroup and heavy ties: the order is python's stable sort."""
    rng = np.random.default_rng(1)
    for n, k, hi in ((1 << 20, 60, 300), ((1 << 20) + 12345, 840, 7), (1500000, 3000, 2)):
        counts = rng.integers(0, hi, n).astype(np.int32)
        ref = np.lexsort((np.arange(n), -counts.astype(np.int64)))[:k]
        np.testing.assert_array_equal(lib.topk(counts, k), ref)


def _refine_case(seed, n_atoms=400):
    coords, names, elems = synth.random_globule(n_atoms, 14.0, seed)
    m = synth.masses(elems)
    grid, x0, y0, z0 = O.structure_to_density(coords, m, 8.0, 1.5)
    grid = np.pad(grid, 6)
    origin = np.array([x0, y0, z0]) - 6 * 1.5
    rng = np.random.default_rng(seed + 100)
    ang = 0.12
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    from mad_amd.math_utils import euler_rod_mat
    R = euler_rod_mat(ax, ang)
    start = (coords - coords.mean(0)) @ R + coords.mean(0) + rng.normal(scale=1.2, size=3)
    return grid, origin, 1.5, coords, start


def test_refine_matches_oracle(lib):
    grid, origin, vs, truth, start = _refine_case(4)
    lib.upload_density(grid, origin, vs)
    for n_steps in (1, 2, 3, 8, 500):
        ref, rconv, rlast, _ = O.refine(grid, origin, vs, start, n_steps=n_steps, max_step=1.0, min_step=0.1)
        got, gconv, glast = lib.refine(start, n_steps=n_steps, max_step=1.0, min_step=0.1)
        assert (gconv, glast) == (rconv, rlast)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7)
    # the refined pose moves towards the planted one
    assert np.sqrt(((got - truth) ** 2).sum(1).mean()) < 0.6 * np.sqrt(((start - truth) ** 2).sum(1).mean())


def test_refine_batch_of_candidates(lib):
    grid, origin, vs, truth, start = _refine_case(5)
    lib.upload_density(grid, origin, vs)
    starts = np.stack([start, start + 0.7, truth + 30.0])      # the last one is mostly outside the map
    got, conv, last = lib.refine(starts, n_steps=500, max_step=1.0, min_step=0.1)
    for i in range(3):
        ref, rconv, rlast, _ = O.refine(grid, origin, vs, starts[i], n_steps=500, max_step=1.0, min_step=0.1)
        assert (bool(conv[i]), int(last[i])) == (rconv, rlast)
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=1e-7)


def test_refine_split_over_workgroups(lib):
    """Large structures in small batches: mad_refine splits each candidate's atoms over several workgroups that meet at every
    reduction (group_reduce: agent-scope hand-off of the partial sums).  Same trajectory as the oracle: converged / last step
    identical, coordinates to 1e-6 A after up to 500 dependent steps (the summation order differs in the last bits)."""
    coords, names, elems = synth.random_globule(9000, 30.0, 6)
    m = synth.masses(elems)
    grid, x0, y0, z0 = O.structure_to_density(coords, m, 8.0, 1.5)
    grid = np.pad(grid, 8)
    origin = np.array([x0, y0, z0]) - 8 * 1.5
    from mad_amd.math_utils import euler_rod_mat
    starts = []
    for ang, sh in ((0.08, (1.0, -0.8, 0.5)), (0.15, (-1.5, 0.3, 1.1)), (0.02, (40.0, 40.0, 40.0))):      # the last one mostly outside the map
        R = euler_rod_mat(np.array([0.3, -0.5, 0.81]) / np.linalg.norm([0.3, -0.5, 0.81]), ang)
        starts.append((coords - coords.mean(0)) @ R + coords.mean(0) + np.array(sh))
    starts = np.stack(starts)
    lib.upload_density(grid, origin, 1.5)
    for n_steps in (3, 500):
        got, conv, last = lib.refine(starts, n_steps=n_steps, max_step=1.0, min_step=0.1)
        for i in range(len(starts)):
            ref, rconv, rlast, _ = O.refine(grid, origin, 1.5, starts[i], n_steps=n_steps, max_step=1.0, min_step=0.1)
            assert (bool(conv[i]), int(last[i])) == (rconv, rlast)
            np.testing.assert_allclose(got[i], ref, rtol=0, atol=1e-6)
    again, conv2, last2 = lib.refine(starts, n_steps=500, max_step=1.0, min_step=0.1)      # reproducible run to run
    np.testing.assert_array_equal(again, got)
    assert np.sqrt(((got[0] - coords) ** 2).sum(1).mean()) < 0.6 * np.sqrt(((starts[0] - coords) ** 2).sum(1).mean())


def test_density_and_ccc_match_oracle(lib):
    coords, names, elems = synth.random_globule(500, 15.0, 8)
    m = synth.masses(elems)
    for res, vs, iso in ((8.0, 1.5, 0.0), (6.0, 1.2, 0.05)):
        ref, rx, ry, rz = O.structure_to_density(coords, m, res, vs, isovalue=iso)
        got, gx, gy, gz = lib.structure_to_density(coords, m, res, vs, isovalue=iso)
        assert got.shape == ref.shape and (gx, gy, gz) == (rx, ry, rz)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)
    # CCC of the map against a shifted copy of the structure (partial overlap, negative voxels clamped)
    big = np.pad(ref, 5).astype(np.float32)
    big[2, 3, 4] = -0.5
    o_big = np.array([rx, ry, rz]) - 5 * vs
    sub, sx, sy, sz = O.structure_to_density(coords + np.array([3.1, -2.2, 1.4]), m, res, vs)
    for shift in ((0, 0, 0), (40 * vs, 0, 0), (1000.0, 0, 0)):
        o2 = np.array([sx, sy, sz]) + np.array(shift)
        a1, b1, a2, b2 = big.copy(), sub.copy(), big.copy(), sub.copy()
        r = O.ccc(a1, o_big, b1, o2, vs)
        g = lib.ccc(a2, o_big, b2, o2, vs)
        if np.isnan(r):
            assert np.isnan(g)
        else:
            assert abs(g - r) <= 1e-9 * max(1.0, abs(r))
        np.testing.assert_array_equal(a2, a1)      # clamped in place, like Dmap.py:160-161
        assert a2[2, 3, 4] == 0.0


def test_dock_refine_score_equals_the_separate_calls_and_the_oracle(lib):
    """mad_dock_refine_score (poses in, scores out: placement, refinement, density simulation and CCC of all candidates on the device)
    against (a) the oracle chain on the host-placed coordinates and (b) mad_refine + mad_density_ccc; candidates of two structures
    of different sizes in one batch; a candidate that leaves the map; the batch reproduces itself bit for bit."""
    from mad_amd.math_utils import euler_rod_mat
    grid, origin, vs, truth, _ = _refine_case(4)
    other, _, elems_o = synth.random_globule(700, 12.0, 9)
    structs = [(truth, synth.masses(synth.random_globule(len(truth), 10.0, 4)[2])), (other + truth.mean(0), synth.masses(elems_o))]
    lib.upload_density(grid, origin, vs)
    rng = np.random.default_rng(12)
    hi_p, lo_p, rot, owner, starts = [], [], [], [], []
    for c, (st, ang, shift) in enumerate(((0, 0.10, 1.0), (0, 0.05, -0.8), (1, 0.2, 0.5), (0, 0.02, 60.0), (1, 0.0, 0.0))):
        atoms = structs[st][0]
        ax = rng.normal(size=3)
        R = euler_rod_mat(ax / np.linalg.norm(ax), ang)
        hi = atoms[rng.integers(len(atoms))] + 0.25
        lo = hi + shift * np.array([0.6, -0.3, 0.5])
        hi_p.append(hi); lo_p.append(lo); rot.append(R); owner.append(st)
        starts.append((atoms - hi) @ R + lo)      # MaD.py:566-569
    res = 8.0
    coords, conv, last, ccc = lib.dock_refine_score([a for a, _ in structs], [m for _, m in structs], np.array(hi_p), np.array(lo_p),
                                                   np.array(rot), res, n_steps=500, max_step=1.0, min_step=0.1, cand_struct=np.array(owner))
    for c, st in enumerate(owner):
        ref, rconv, rlast, _ = O.refine(grid, origin, vs, starts[c], n_steps=500, max_step=1.0, min_step=0.1)
        assert (bool(conv[c]), int(last[c])) == (rconv, rlast)
        np.testing.assert_allclose(coords[c], ref, rtol=0, atol=1e-6)
        g2, x0, y0, z0 = O.structure_to_density(ref, structs[st][1], res, vs)
        want = O.ccc(grid.copy(), origin, g2, np.array([x0, y0, z0]), vs)
        if np.isnan(want):      # a box that misses every occupied voxel of the map: 0 / 0, as the reference (Dmap.py:249-254)
            assert np.isnan(ccc[c]), (c, ccc[c])
        else:
            assert abs(ccc[c] - want) <= 1e-5 * max(abs(want), 1e-3), (c, ccc[c], want)
    # the separate calls on one structure's candidates
    sel = [c for c, st in enumerate(owner) if st == 0]
    got, gconv, glast = lib.refine(np.stack([starts[c] for c in sel]), n_steps=500, max_step=1.0, min_step=0.1)
    for j, c in enumerate(sel):
        np.testing.assert_allclose(coords[c], got[j], rtol=0, atol=1e-6)
        assert (bool(gconv[j]), int(glast[j])) == (bool(conv[c]), int(last[c]))
    np.testing.assert_allclose(lib.density_ccc(got, structs[0][1], res), ccc[sel], rtol=1e-6, atol=1e-9, equal_nan=True)
    # single-structure form, coordinates left on the device; and run-to-run reproducibility of the whole batch
    _, c1, l1, ccc1 = lib.dock_refine_score(structs[0][0], structs[0][1], np.array(hi_p)[sel], np.array(lo_p)[sel], np.array(rot)[sel], res,
                                            n_steps=500, max_step=1.0, min_step=0.1, want_coords=False)
    np.testing.assert_allclose(ccc1, ccc[sel], rtol=1e-9, atol=0, equal_nan=True)
    coords2, conv2, last2, ccc2 = lib.dock_refine_score([a for a, _ in structs], [m for _, m in structs], np.array(hi_p), np.array(lo_p),
                                                       np.array(rot), res, n_steps=500, max_step=1.0, min_step=0.1, cand_struct=np.array(owner))
    np.testing.assert_array_equal(ccc2, ccc)
    for a, b in zip(coords, coords2):
        np.testing.assert_array_equal(a, b)


def test_set_pipeline_and_batched_match(lib, fields):
    """The device-resident pipeline (set_build + match_topk / match_topk_many) equals the stage-by-stage oracle chain."""
    sets, host = [], []
    f = fields[1]
    for seed, n in ((31, 50), (32, 18), (33, 22)):
        coords = synth.interior_anchors(f["shape"], n, 10, seed)
        subv = coords.astype(np.float64) * 1.5 + 0.37
        sets.append(lib.set_build([-1, f["slot"]], coords, np.ones(n, np.int32), subv, np.arange(n)))
        rows = O.orient(f["gx"], f["gy"], f["gz"], 1, coords, E112.sphere_eqsp, E112.p_centers_eqsp, want_counts=False)
        dsc = O.describe(f["gx"], f["gy"], f["gz"], 1, coords[rows["anchor"]], rows["R"], E16.sphere_eqsp)
        host.append(dict(rows=rows, dsc=dsc, subv=subv))
        d = sets[-1].download()
        np.testing.assert_array_equal(d["dsc"], dsc)
        np.testing.assert_array_equal(d["anchor"], rows["anchor"])
        np.testing.assert_array_equal(d["main"], rows["main"])
        np.testing.assert_array_equal(d["sec"], rows["sec"])
    lo, his = sets[0], sets[1:]
    cc, k = 0.3, 40
    batched = lib.match_topk_many(his, lo, cc, 4.0, k)
    for hi, hh, (btop, bidx, bst) in zip(his, host[1:], batched):
        top, idx, st = lib.match_topk(hi, lo, cc, 4.0, k)
        np.testing.assert_array_equal(btop, top)
        np.testing.assert_array_equal(bidx, idx)
        assert bst == st
        ph, pl, ps, _ = O.correlate(hh["dsc"], host[0]["dsc"], cc)
        assert st["n_pairs"] == len(ph) > 100
        hi_p, lo_p = hh["subv"][hh["rows"]["anchor"]], host[0]["subv"][host[0]["rows"]["anchor"]]
        mh = np.stack([hh["rows"]["anchor"], np.ones_like(hh["rows"]["anchor"]), hh["rows"]["main"]], 1)
        ml = np.stack([host[0]["rows"]["anchor"], np.ones_like(host[0]["rows"]["anchor"]), host[0]["rows"]["main"]], 1)
        res, cnt = O.pose_score(ph, pl, ps, hi_p, hh["rows"]["R"], mh, lo_p, host[0]["rows"]["R"], ml,
                                np.unique(hi_p[np.unique(ph)], axis=0), np.unique(lo_p[np.unique(pl)], axis=0), 4.0)
        order = O.topk(cnt, k)
        np.testing.assert_array_equal(idx, order)
        np.testing.assert_allclose(top, res[order], rtol=1e-10, atol=1e-10)
    # rebuilding a set in place (the describe launch is then sized from the previous row count) changes nothing,
    # also when the new anchor list yields MORE rows than the hint
    small = synth.interior_anchors(f["shape"], 6, 10, 77)
    s2 = lib.set_build([-1, f["slot"]], small, np.ones(6, np.int32), small.astype(np.float64), np.arange(6))
    n_small, _ = s2.size()
    big = synth.interior_anchors(f["shape"], 60, 10, 78)
    lib.set_build([-1, f["slot"]], big, np.ones(60, np.int32), big.astype(np.float64), np.arange(60), into=s2)
    rows = O.orient(f["gx"], f["gy"], f["gz"], 1, big, E112.sphere_eqsp, E112.p_centers_eqsp, want_counts=False)
    dsc = O.describe(f["gx"], f["gy"], f["gz"], 1, big[rows["anchor"]], rows["R"], E16.sphere_eqsp)
    assert len(dsc) > n_small + n_small // 8 + 64
    np.testing.assert_array_equal(s2.download()["dsc"], dsc)


def test_anchors_with_identical_coordinates_count_once(lib, fields):
    """Two detector peaks that converge on the same sub-voxel position are ONE point of a cloud: the reference builds the
    clouds with np.unique(coords, axis=0) (MaD.py:427-428).  Sets built on the device get the same clouds (and sizes l)."""
    f = fields[1]
    built, host = [], []
    for seed, n in ((41, 40), (42, 24)):
        coords = synth.interior_anchors(f["shape"], n, 10, seed)
        subv = coords.astype(np.float64) * 1.5 + 0.37
        # the last quarter repeats the sub-voxel position (and voxel) of the first quarter
        q = n // 4
        coords[-q:], subv[-q:] = coords[:q], subv[:q]
        built.append(lib.set_build([-1, f["slot"]], coords, np.ones(n, np.int32), subv, np.arange(n)))
        rows = O.orient(f["gx"], f["gy"], f["gz"], 1, coords, E112.sphere_eqsp, E112.p_centers_eqsp, want_counts=False)
        dsc = O.describe(f["gx"], f["gy"], f["gz"], 1, coords[rows["anchor"]], rows["R"], E16.sphere_eqsp)
        host.append(dict(rows=rows, dsc=dsc, subv=subv))
    (lo, hi), (hl, hh) = built, host
    cc, k = 0.3, 30
    top, idx, st = lib.match_topk(hi, lo, cc, 4.0, k)
    ph, pl, ps, _ = O.correlate(hh["dsc"], hl["dsc"], cc)
    assert st["n_pairs"] == len(ph) > 50
    hi_p, lo_p = hh["subv"][hh["rows"]["anchor"]], hl["subv"][hl["rows"]["anchor"]]
    hi_cloud, lo_cloud = np.unique(hi_p[np.unique(ph)], axis=0), np.unique(lo_p[np.unique(pl)], axis=0)
    assert len(hi_cloud) < len(np.unique(hh["rows"]["anchor"][np.unique(ph)]))      # the case does hold duplicates
    assert (st["l_hi"], st["l_lo"]) == (len(hi_cloud), len(lo_cloud))
    mh = np.stack([hh["rows"]["anchor"], np.ones_like(hh["rows"]["anchor"]), hh["rows"]["main"]], 1)
    ml = np.stack([hl["rows"]["anchor"], np.ones_like(hl["rows"]["anchor"]), hl["rows"]["main"]], 1)
    res, cnt = O.pose_score(ph, pl, ps, hi_p, hh["rows"]["R"], mh, lo_p, hl["rows"]["R"], ml, hi_cloud, lo_cloud, 4.0)
    np.testing.assert_array_equal(lib.match_fetch(st["n_pairs"])[3], cnt)
    order = O.topk(cnt, k)
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_allclose(top, res[order], rtol=1e-10, atol=1e-10)


def test_counts_of_a_pruned_match_are_refused_once_a_set_is_gone(lib, fields):
    """mad_match_topk prunes its pose search, so mad_match_fetch(counts) / mad_match_results have to search the other pairs
    afterwards -- through the two sets.  A set destroyed or rebuilt in between must be refused, not dereferenced."""
    from mad_amd._lib import MadBackendError
    f = fields[1]
    sets = []
    for seed, n in ((51, 48), (52, 20)):
        coords = synth.interior_anchors(f["shape"], n, 10, seed)
        sets.append((coords, lib.set_build([-1, f["slot"]], coords, np.ones(n, np.int32), coords.astype(np.float64) * 1.5 + 0.2, np.arange(n))))
    (lo_c, lo), (hi_c, hi) = sets
    top, idx, st = lib.match_topk(hi, lo, 0.3, 4.0, 5)
    assert st["n_pairs"] > 50 and lib.last_pose_selected() < st["n_pairs"]      # the search was pruned
    full = lib.match_fetch(st["n_pairs"])[3]                                    # completes the counts: fine, both sets alive
    assert len(full) == st["n_pairs"]
    lib.match_topk(hi, lo, 0.3, 4.0, 5)
    lib.set_build([-1, f["slot"]], hi_c, np.ones(len(hi_c), np.int32), hi_c.astype(np.float64) * 1.5 + 0.2, np.arange(len(hi_c)), into=hi)
    with pytest.raises(MadBackendError, match="rebuilt"):
        lib.match_fetch(st["n_pairs"])
    lib.match_topk(hi, lo, 0.3, 4.0, 5)
    hi.close()
    with pytest.raises(MadBackendError, match="destroyed"):
        lib.match_fetch(st["n_pairs"])
    lo.close()


def test_lanes_overlap_and_serial_give_identical_results(lib):
    """Builds and matches of several structures spread over the lanes (streams); serialising the lanes with
    mad_set_overlap(0) must not change a bit of the output."""
    from mad_amd._lib import DeviceSet
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    rng = np.random.default_rng(11)
    vol = synth.blob_volume((72, 70, 68), n_blobs=60, seed=5, sigma=(1.5, 3.5))
    slot = lib.new_slot()
    lib.upload_field(slot, synth.gradient_field(vol))
    structs = []
    for s_ in range(4):
        n = 50 + 10 * s_
        coords = np.stack([rng.integers(12, d - 12, n) for d in vol.shape], 1).astype(np.int32)
        structs.append((coords, np.ones(n, np.int32), coords.astype(np.float64) * 1.5 + rng.normal(scale=0.2, size=(n, 3)), np.arange(n, dtype=np.int32)))
    outs = []
    for overlap in (True, False, True):
        lib.set_overlap(overlap)
        sets = [lib.set_build([-1, slot], *st) for st in structs]
        res = lib.match_topk_many(sets[1:], sets[0], 0.5, 4.0, 25)
        outs.append([(top.copy(), idx.copy(), dict(st)) for top, idx, st in res])
        rows = [s_.download() for s_ in sets]
        outs[-1].append(rows)
        for s_ in sets:
            s_.close()
    lib.set_overlap(True)
    lib.free_field(slot)
    for other in outs[1:]:
        for a, b in zip(outs[0][:-1], other[:-1]):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            assert a[2] == b[2]
        for ra, rb in zip(outs[0][-1], other[-1]):
            for key in ("anchor", "main", "sec", "R", "dsc"):
                np.testing.assert_array_equal(ra[key], rb[key])
    assert any(len(o[0]) for o in outs[0][:-1])


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_match_equals_unsharded(lib, world):
    """One subunit's pair grid split into `world` blocks of map rows (mad_match_shard_pairs / _topk + the host merge of
    mad_amd.dist): flags OR-ed, per-shard top-k merged by global pair rank -- the unsharded result, rows and order."""
    from mad_amd import dist as mdist
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    shape = (56, 60, 64)
    vol = synth.blob_volume(shape, n_blobs=60, seed=9, sigma=(1.5, 3.5))
    slot = lib.new_slot()
    lib.upload_field(slot, synth.gradient_field(vol))
    rng = np.random.default_rng(world)
    sets = []
    for n in (150, 60):
        coords = synth.interior_anchors(shape, n, 12, 100 + n)
        subv = coords.astype(np.float64) * 1.5 + rng.normal(scale=0.2, size=(n, 3))
        sets.append(lib.set_build([-1, slot], coords, np.ones(n, np.int32), subv, np.arange(n)))
    lo, hi = sets
    cc, dist_, k = 0.45, 4.0, 40
    top, idx, st = lib.match_topk(hi, lo, cc, dist_, k)
    assert st["n_pairs"] > 200 and len(top) == k
    ph, pl, _, cnt = lib.match_fetch(st["n_pairs"])
    n_lo = lo.size()[0]
    ref_rank = ph[idx].astype(np.int64) * n_lo + pl[idx]
    # every "rank" in turn on this one GPU: stage B of all shards first (Exchange 1 needs all flags) ...
    flags = []
    for r in range(world):
        b, e = mdist.lo_row_block(n_lo, r, world)
        uh, ul, _ = lib.match_shard_pairs(hi, lo, b, e, cc)
        flags.append(np.concatenate([uh, ul]))
    flags_all = np.bitwise_or.reduce(np.stack(flags), axis=0)
    # ... then stages B + C per shard with the reduced flags, and Exchange 2 as a plain list
    shards = []
    for r in range(world):
        shards.append(mdist.sharded_match(lib, hi, lo, cc, dist_, k, r, world, reduce_flags=lambda f: flags_all, gather=lambda parts: [parts])[0:3])
    rows, counts, ranks = mdist.merge_topk([s[0] for s in shards], [s[1] for s in shards], [s[2] for s in shards], k)
    np.testing.assert_array_equal(ranks, ref_rank)
    np.testing.assert_array_equal(counts, cnt[idx])
    np.testing.assert_array_equal(rows, top)
    for s_ in sets:
        s_.close()
    lib.free_field(slot)


@pytest.mark.parametrize("n_sets", [1, 5, 19])
def test_sets_built_in_one_batch_equal_sets_built_one_by_one(lib, fields, n_sets):
    """mad_set_build_many: the anchors of several structures (different fields, one or both octaves, an empty list, border
    rejects) through ONE k_orient / scan / row-expansion / k_describe launch each: every set is bit for bit the set mad_set_build
    makes on its own, also when the batch is rebuilt in place with other anchors (stale launch-size hints) and when there are
    more structures than one batch of kernel arguments holds (19 > 16)."""
    f0, f1 = fields[0], fields[1]
    rng = np.random.default_rng(40 + n_sets)

    def job(i, shrink=0):
        kind = i % 4
        a0 = _anchors(f0["shape"], 0, 24 + 3 * (i % 5) - shrink, 100 + i)
        a1 = _anchors(f1["shape"], 1, 30 + 2 * (i % 7) - shrink, 200 + i)
        if kind == 1:
            a0 = a0[:0]
        if kind == 2:
            a1 = a1[:0]
        if kind == 3 and i > 3:
            a0, a1 = a0[:0], a1[:0]      # a structure without anchors
        coords = np.concatenate([a0, a1]).astype(np.int32)
        octave = np.concatenate([np.zeros(len(a0), np.int32), np.ones(len(a1), np.int32)])
        subv = coords * np.where(octave[:, None] == 0, 0.75, 1.5) + rng.normal(scale=0.2, size=coords.shape)
        slots = [f0["slot"] if kind != 1 else -1, f1["slot"] if kind != 2 else -1]
        return (slots, coords, octave, subv, np.arange(len(coords), dtype=np.int32) + 10 * i)

    jobs = [job(i) for i in range(n_sets)]
    batch = lib.set_build_many(jobs)
    assert len(batch) == n_sets
    total = 0
    for i, (j, got) in enumerate(zip(jobs, batch)):
        one = lib.set_build(*j)
        assert got.size() == one.size(), "set %d" % i
        total += one.size()[0]
        a, b = got.download(), one.download()
        for key in ("anchor", "main", "sec", "R", "dsc"):
            np.testing.assert_array_equal(a[key], b[key], err_msg="set %d: %s" % (i, key))
        one.close()
    assert total > 40 * n_sets
    # the same sets rebuilt in place, with fewer / other anchors (and set 0 matched against set 1 afterwards)
    jobs2 = [job(i + 1, shrink=6) for i in range(n_sets)]
    again = lib.set_build_many([j + (d,) for j, d in zip(jobs2, batch)])
    for i, (j, got) in enumerate(zip(jobs2, again)):
        one = lib.set_build(*j)
        assert got.size() == one.size(), "rebuilt set %d" % i
        a, b = got.download(), one.download()
        for key in ("anchor", "main", "sec", "R", "dsc"):
            np.testing.assert_array_equal(a[key], b[key], err_msg="rebuilt set %d: %s" % (i, key))
        if i == 1 and n_sets > 1:
            top_a = lib.match_topk(again[0], got, 0.3, 4.0, 20)
            ref0 = lib.set_build(*jobs2[0])
            top_b = lib.match_topk(ref0, one, 0.3, 4.0, 20)
            np.testing.assert_array_equal(top_a[0], top_b[0])
            np.testing.assert_array_equal(top_a[1], top_b[1])
            assert top_a[2] == top_b[2]
            ref0.close()
        one.close()
    for s_ in again:
        s_.close()


@pytest.mark.parametrize("r,lim_main,lim_sec", [(4, 6, 6), (10, 6, 6), (8, 3, 4), (8, 1, 1)])
def test_orient_other_box_sizes_and_limits_match_oracle(lib, fields, r, lim_main, lim_sec):
    """`Orientator(ori_radius = 8 | 20)` (box side r = 4 / 10: the 10-voxel sphere holds 4 945 voxels, two request batches per
    thread and 77 KB of LDS) and other `main_ori` / `sec_ori` limits than 6 / 6 (Orientator.py:13, 181-184, 232-235)."""
    for octave in (1, 0):
        f = fields[octave]
        margin = r * (1 if octave == 1 else 2) + 1
        coords = np.concatenate([synth.interior_anchors(f["shape"], 30, margin + 1, 90 + r + octave),
                                 np.array([[margin - 2, f["shape"][1] // 2, f["shape"][2] // 2]], np.int32)]).astype(np.int32)      # one border reject
        ref = O.orient(f["gx"], f["gy"], f["gz"], octave, coords, E112.sphere_eqsp, E112.p_centers_eqsp, r=r, lim_main=lim_main, lim_sec=lim_sec)
        got = lib.orient(f["slot"], octave, coords, r=r, lim_main=lim_main, lim_sec=lim_sec)
        assert got["n_reject"] == ref["n_reject"] >= 1
        if (lim_main, lim_sec) == (6, 6):
            assert len(ref["anchor"]) > 10
        for key in ("anchor", "main", "sec", "counts"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=key)
        np.testing.assert_allclose(got["R"], ref["R"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("r", [2, 4, 6])
def test_describe_other_lattice_sizes_match_oracle(lib, fields, r):
    """`Descriptor(dsc_radius = 4 | 8 | 12)` (Descriptor.py:34-35: 2 r samples per axis): k_describe<4>, <8>, <12>.  Their last
    classification chunk is short (4 of 8 samples) -- a round-2 review found it reading four texels past the thread's array."""
    for octave in (1, 0):
        f = fields[octave]
        coords = synth.interior_anchors(f["shape"], 40, (8 if octave == 1 else 16) + 2, 70 + r)
        rows = O.orient(f["gx"], f["gy"], f["gz"], octave, coords, E112.sphere_eqsp, E112.p_centers_eqsp)
        rc, R = coords[rows["anchor"]], rows["R"]
        assert len(rc) > 40
        ref = O.describe(f["gx"], f["gy"], f["gz"], octave, rc, R, E16.sphere_eqsp, r=r)
        got = lib.describe(f["slot"], octave, rc, R, r=r)
        assert ref.sum() > 0 and ref.sum() == got.sum()
        np.testing.assert_array_equal(got, ref)


def test_matches_of_a_bracket_with_one_gemm_grid(lib, fields):
    """mad_set_batching: the score tiles of all matches of a bracket from ONE grid (jobs of different sizes, a subunit without
    rows, two subunits on one lane, i.e. one of them outside the batch): top-k rows, pair ranks and statistics are those of the
    one-GEMM-per-match bracket, bit for bit."""
    f0, f1 = fields[0], fields[1]
    a0, a1 = _anchors(f0["shape"], 0, 36, 31), _anchors(f1["shape"], 1, 44, 32)
    coords = np.concatenate([a0, a1]).astype(np.int32)
    octave = np.concatenate([np.zeros(len(a0), np.int32), np.ones(len(a1), np.int32)])
    lo = lib.set_build([f0["slot"], f1["slot"]], coords, octave, coords * np.where(octave[:, None] == 0, 0.75, 1.5), np.arange(len(coords), dtype=np.int32))
    his = []
    for i, n in enumerate((30, 12, 0, 41, 25, 18, 33, 27, 22, 16)):      # ten subunits on eight lanes
        c = _anchors(f1["shape"], 1, n, 60 + i)[:n] if n else np.zeros((0, 3), np.int32)
        his.append(lib.set_build([-1, f1["slot"]], c, np.ones(len(c), np.int32), c * 1.5 + 0.1 * i, np.arange(len(c), dtype=np.int32)))
    assert len({h.lane() for h in his}) < len(his)
    lib.set_batching(False)
    want = lib.match_topk_many(his, lo, 0.3, 4.0, 25)
    lib.set_batching(True)
    try:
        got = lib.match_topk_many(his, lo, 0.3, 4.0, 25)
        again = lib.match_topk_many(his[:3], lo, 0.3, 4.0, 25)
    finally:
        lib.set_batching(False)
    assert sum(w[2]["n_pairs"] for w in want) > 200
    for (ta, ia, sa), (tb, ib, sb) in list(zip(got, want)) + list(zip(again, want[:3])):
        assert sa == sb
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(ta, tb)
    for h in his:
        h.close()
    lo.close()


@pytest.mark.parametrize("n_shares", [1, 2, 3, 8])
def test_set_built_in_shares_equals_unsharded(lib, fields, n_shares):
    """SURVEY.md 8(e) stage A on one GPU: the anchors of a structure (both octaves, border rejects included) dealt round-robin
    to n_shares shares, each built on its own (mad_set_build), exported as a wire image and imported into one set
    (mad_set_import): rows, bins, Rfinal, descriptors and the match against it are those of the set built from the whole list,
    bit for bit.  Host images and device images (what the RCCL all-gather moves) both."""
    import torch
    from mad_amd import dist as mdist
    f0, f1 = fields[0], fields[1]
    a0, a1 = _anchors(f0["shape"], 0, 40, 21), _anchors(f1["shape"], 1, 50, 22)
    coords = np.concatenate([a0, a1]).astype(np.int32)
    octave = np.concatenate([np.zeros(len(a0), np.int32), np.ones(len(a1), np.int32)])
    rng = np.random.default_rng(5)
    subv = coords * np.where(octave[:, None] == 0, 0.75, 1.5) + rng.normal(scale=0.2, size=coords.shape)
    subv[7] = subv[3]      # two anchors on one position: one point of a cloud (anc_canon must survive the import)
    index = np.arange(len(coords), dtype=np.int32) + 100
    slots = [f0["slot"], f1["slot"]]
    ref = lib.set_build(slots, coords, octave, subv, index)
    want = ref.download()
    n_ref = ref.size()[0]
    assert n_ref > 150 and len(np.unique(want["anchor"])) < len(coords)      # some anchors are rejected
    shares = []
    for r in range(n_shares):
        sel = mdist.share_of(len(coords), r, n_shares)
        shares.append(lib.set_build(slots, coords[sel], octave[sel], subv[sel], index[sel]))
    cap = max(s.size()[0] for s in shares) + 3
    nbytes = lib.set_wire_bytes(cap)
    host = np.concatenate([lib.set_export(s, cap) for s in shares])
    dev = torch.zeros(n_shares * nbytes, dtype=torch.uint8, device="cuda")
    for r, s in enumerate(shares):
        lib.set_export(s, cap, device_ptr=dev.data_ptr() + r * nbytes)
    lib.synchronize()      # (the images agree up to the unused tail of every section, which is not defined)
    # a subunit to match against the assembled set
    hi_c = _anchors(f1["shape"], 1, 30, 23)
    hi = lib.set_build([-1, f1["slot"]], hi_c, np.ones(len(hi_c), np.int32), hi_c * 1.5 + 0.1, np.arange(len(hi_c)))
    top_ref, idx_ref, st_ref = lib.match_topk(hi, ref, 0.3, 4.0, 30)
    assert st_ref["n_pairs"] > 50
    for kind in ("host", "device"):
        if kind == "host":
            full = lib.set_import(n_shares, cap, coords, octave, subv, index, wires=host)
        else:
            full = lib.set_import(n_shares, cap, None, octave, subv, index, device_ptr=dev.data_ptr())
        assert full.size() == (n_ref, len(coords))
        got = full.download()
        for key in ("anchor", "main", "sec", "R", "dsc"):
            np.testing.assert_array_equal(got[key], want[key], err_msg="%s (%s images)" % (key, kind))
        top, idx, st = lib.match_topk(hi, full, 0.3, 4.0, 30)
        assert st == st_ref
        np.testing.assert_array_equal(idx, idx_ref)
        np.testing.assert_array_equal(top, top_ref)
        # rebuilt in place from the same images: still the same set
        full = lib.set_import(n_shares, cap, coords, octave, subv, index, wires=host, into=full)
        np.testing.assert_array_equal(full.download()["dsc"], want["dsc"])
        full.close()
    # images too small for the largest share: the import says so when the set is first used
    small = max(s.size()[0] for s in shares) - 1
    from mad_amd._lib import MadBackendError
    bad = lib.set_import(n_shares, small, coords, octave, subv, index, wires=np.concatenate([lib.set_export(s, small) for s in shares]))
    with pytest.raises(MadBackendError, match="ENOSPC"):
        bad.size()
    host2 = host.copy()
    host2[28:32] = 0      # magic word of share 0
    bad = lib.set_import(n_shares, cap, coords, octave, subv, index, wires=host2, into=bad)
    with pytest.raises(MadBackendError, match="EINVAL"):
        bad.size()
    for s in shares + [ref, hi, bad]:
        s.close()


def test_sharded_set_build_object_on_one_rank(lib, fields):
    """mad_amd.dist.ShardedSetBuild in rehearsal mode (rank 0 of 4 on one GPU, the other shares built once): every call
    returns the set of the whole anchor list, through device images and the library's own stream."""
    from mad_amd import dist as mdist
    f1 = fields[1]
    coords = _anchors(f1["shape"], 1, 60, 31)
    n = len(coords)
    octave, subv, index = np.ones(n, np.int32), coords * 1.5 + 0.25, np.arange(n, dtype=np.int32)
    ref = lib.set_build([-1, f1["slot"]], coords, octave, subv, index)
    want = ref.download()
    b = mdist.ShardedSetBuild(lib, [-1, f1["slot"]], coords, octave, subv, index, 0, 1, emulate=(4, 0))
    for overlap in (True, False, True):
        lib.set_overlap(overlap)
        for _ in range(2):
            got = b.enqueue().download()
            for key in ("anchor", "main", "sec", "R", "dsc"):
                np.testing.assert_array_equal(got[key], want[key])
    assert b.share.size()[0] < ref.size()[0]
    # the share built in one batch with another structure (bench.py --batched: build_job() + set_build_many + finish())
    other = _anchors(f1["shape"], 1, 25, 32)
    jobs = [b.build_job(), ([-1, f1["slot"]], other, np.ones(len(other), np.int32), other * 1.5, np.arange(len(other), dtype=np.int32))]
    batch = lib.prepare_build_many(jobs)
    for _ in range(2):
        built = batch.run()
        got = b.finish().download()
        for key in ("anchor", "main", "sec", "R", "dsc"):
            np.testing.assert_array_equal(got[key], want[key])
    alone = lib.set_build(*jobs[1])
    np.testing.assert_array_equal(built[1].download()["dsc"], alone.download()["dsc"])
    alone.close()
    built[1].close()
    b.close()
    ref.close()


@pytest.mark.parametrize("k,n_hi_a", [(1, 90), (7, 90), (60, 90), (400, 90), (5000, 90), (60, 600), (60, 900)])
def test_pruned_pose_search_returns_the_rows_of_the_full_search(lib, k, n_hi_a):
    """mad_match_topk only reports k pairs, so the pose search brackets every pair's count with the occupancy bitmaps and
    searches exactly only the pairs whose upper bound reaches the k-th largest lower bound.  On clouds made to produce ties
    en masse at the k-th place (a planted pose that a third of the rows share, rows duplicated, lo points exactly at the
    distance threshold) the k rows and their order must be those of python's stable sort over the EXACT counts of all pairs
    (MaD.py:480), which mad_match_fetch delivers afterwards."""
    rng = np.random.default_rng(11)
    n_lo_a = 260      # (n_hi_a = 600 / 900: hi clouds of 10 / 15 sets of 64 points -- k_pose_bounds<10>, <16> -- and the two-launch form by itself)
    lo_p = rng.uniform(0, 110, size=(n_lo_a, 3))
    Q = synth.random_rotation(rng)
    shift = np.array([7.0, -3.0, 5.0])
    hi_p = np.zeros((n_hi_a, 3))
    # 60 hi anchors are rigid images of lo anchors (the planted pose), half of them displaced by exactly 4 A or a hair less / more
    src = rng.choice(n_lo_a, 60, replace=False)
    d = np.zeros((60, 3))
    d[20:40, 0] = 4.0
    d[40:50, 1] = 4.0 - 1e-9
    d[50:60, 2] = 4.0 + 1e-9
    hi_p[:60] = (lo_p[src] + d - shift) @ Q      # so that (x - p_hi) @ R.T + p_lo with R = Q.T-ish brings them back
    hi_p[60:] = rng.uniform(0, 110, size=(n_hi_a - 60, 3))
    D = 1024
    base = rng.integers(0, 12, size=(8, D)).astype(np.int16)      # 8 descriptor prototypes: rows of the same one correlate perfectly

    def rows_for(n_anchor, per_anchor, seed, planted_R):
        r = np.random.default_rng(seed)
        anchor = np.repeat(np.arange(n_anchor), per_anchor)
        n = len(anchor)
        R = np.stack([synth.random_rotation(r) for _ in range(n)])
        R[::3] = planted_R      # every third row carries the planted frame: pairs of such rows give the same pose
        dsc = base[r.integers(0, 8, size=n)]
        return anchor.astype(np.int32), R, dsc

    la, lR, ldsc = rows_for(n_lo_a, 3, 1, np.eye(3))
    ha, hR, hdsc = rows_for(n_hi_a, 3, 2, Q.T)
    lo = lib.set_load(la, np.zeros(len(la), np.int32), lR, ldsc, lo_p, np.arange(n_lo_a), np.ones(n_lo_a, np.int32))
    hi = lib.set_load(ha, np.zeros(len(ha), np.int32), hR, hdsc, hi_p, np.arange(n_hi_a), np.ones(n_hi_a, np.int32))
    top, idx, st = lib.match_topk(hi, lo, 0.9, 4.0, k)
    n_sel = lib.last_pose_selected()
    assert st["n_pairs"] > 20000
    ph, pl, ps, cnt = lib.match_fetch(st["n_pairs"])      # exact counts of ALL pairs (completes what the pruning skipped)
    order = np.lexsort((np.arange(len(cnt)), -cnt.astype(np.int64)))[:k]
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_array_equal(top[:, 1], 100.0 * cnt[order] / st["l_hi"])
    assert len(np.unique(cnt[order])) < len(order) or k == 1      # ties inside the selection ...
    if k < len(cnt):
        assert cnt[order[-1]] == np.sort(cnt)[::-1][k - 1]
    if k <= 400:
        assert np.sum(cnt == cnt[order[-1]]) > np.sum(cnt[order] == cnt[order[-1]]) or k == 1      # ... and across its edge
        assert 0 < n_sel < st["n_pairs"]      # the bounds did exclude pairs
    # the same call again (hints warmed, a pruned match behind it) and the all-pairs rows of the drop-in _match_dsc
    top2, idx2, _ = lib.match_topk(hi, lo, 0.9, 4.0, k)
    np.testing.assert_array_equal(idx2, idx)
    np.testing.assert_array_equal(top2, top)
    res = lib.match_results(hi, lo, st["n_pairs"])
    np.testing.assert_array_equal(res[:, 1], 100.0 * cnt / st["l_hi"])
    np.testing.assert_array_equal(res[order], top)
    # the bounds pass looks at the best-scoring pairs first and abandons the others early: with other sizes of that first list
    # (64 k pairs at least; everything in it, i.e. no abandoning at all) the k rows must not change
    for split_min in (0, 4096, 10 ** 9):
        lib.set_option("pose_split", 1)
        lib.set_option("pose_split_min", split_min)
        try:
            top3, idx3, st3 = lib.match_topk(hi, lo, 0.9, 4.0, k)
        finally:
            lib.set_option("pose_split_min", 4096)
            lib.set_option("pose_split", -1)
        assert st3 == st
        np.testing.assert_array_equal(idx3, idx)
        np.testing.assert_array_equal(top3, top)
    hi.close()
    lo.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ori_cap, dsc_cap", [(0, 1 << 20), (3, 1 << 20), (1 << 20, 0), (1 << 20, 5), (2, 7)])
def test_full_queues_of_undecided_directions_change_nothing(lib, fields, ori_cap, dsc_cap):
    """k_orient hands the directions its float32 classifier cannot decide to the exact one through a queue of 512 entries and
    classifies in place beyond; k_describe's table tier queues 768 texel indices and redoes the whole row with the exact arithmetic
    beyond.  With the queues cut to a few entries (mad_set_option "ori_queue" / "dsc_queue") those paths run for every anchor and
    row, and the set comes out bit for bit as with the full queues."""
    f0, f1 = fields[0], fields[1]
    a0 = _anchors(f0["shape"], 0, 40, 71)
    a1 = _anchors(f1["shape"], 1, 50, 72)
    coords = np.concatenate([a0, a1]).astype(np.int32)
    octave = np.concatenate([np.zeros(len(a0), np.int32), np.ones(len(a1), np.int32)])
    subv = coords * np.where(octave[:, None] == 0, 0.75, 1.5)
    job = ([f0["slot"], f1["slot"]], coords, octave, subv, np.arange(len(coords), dtype=np.int32))
    ref = lib.set_build(*job)
    want = ref.download()
    assert ref.size()[0] > 100
    try:
        lib.set_option("ori_queue", ori_cap)
        lib.set_option("dsc_queue", dsc_cap)
        got = lib.set_build(*job)
        have = got.download()
    finally:
        lib.set_option("ori_queue", 1 << 20)
        lib.set_option("dsc_queue", 1 << 20)
    assert got.size() == ref.size()
    for key in ("anchor", "main", "sec", "R", "dsc"):
        np.testing.assert_array_equal(have[key], want[key], err_msg=key)
    got.close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dsc_cap", [1 << 20, 3])
def test_base_octave_rows_from_a_ball_in_lds_equal_rows_described_one_by_one(lib, dsc_cap):
    """k_describe_ball (round 4; mad_set_option "dsc_ball" = 1, off by default): the base-octave anchors whose ball of samples lies
    inside the grid are described anchor by anchor from 11 027 4-byte texels staged in LDS, the others (octave 0, anchors near the
    border) row by row by k_describe.  The set is bit for bit the set k_describe alone makes, and its descriptors are the oracle's
    (Descriptor.py:123-202) -- anchors with 1 .. 20+ rows (several runs of rows per anchor), border anchors, rejected anchors, both
    octaves in one set, rows turned about z only (every sample next to a nearest-voxel tie), and k_describe's full-queue path
    beside it (dsc_cap = 3)."""
    shape1, shape0 = (64, 64, 60), (60, 64, 70)
    _, gx1, gy1, gz1 = _field(shape1, 31)
    _, gx0, gy0, gz0 = _field(shape0, 32)
    s1, s0 = lib.new_slot(), lib.new_slot()
    lib.upload_field(s1, np.stack([gx1, gy1, gz1]))
    lib.upload_field(s0, np.stack([gx0, gy0, gz0]))
    a1 = _anchors(shape1, 1, 70, 73)      # interior ones (ball inside the grid), ones within 14 voxels of the border, rejected ones
    a0 = _anchors(shape0, 0, 12, 74)
    coords = np.concatenate([a1[:30], a0, a1[30:]]).astype(np.int32)
    octave = np.concatenate([np.ones(30, np.int32), np.zeros(len(a0), np.int32), np.ones(len(a1) - 30, np.int32)])
    subv = coords * np.where(octave[:, None] == 0, 0.75, 1.5)
    job = ([s0, s1], coords, octave, subv, np.arange(len(coords), dtype=np.int32))
    inside = np.all((a1 - 14 >= 1) & (a1 + 14 <= np.array(shape1) - 2), axis=1)
    assert inside.sum() >= 20 and (~inside).sum() >= 10, "the anchors must exercise both kernels"
    try:
        lib.set_option("dsc_queue", dsc_cap)
        lib.set_option("dsc_ball", 1)
        ball = lib.set_build(*job)
        have = ball.download()
        lib.set_option("dsc_ball", 0)
        rowwise = lib.set_build(*job)
        want = rowwise.download()
    finally:
        lib.set_option("dsc_ball", 0)
        lib.set_option("dsc_queue", 1 << 20)
    assert ball.size() == rowwise.size() and ball.size()[0] > 200
    for key in ("anchor", "main", "sec", "R", "dsc"):
        np.testing.assert_array_equal(have[key], want[key], err_msg=key)
    fan = np.bincount(have["anchor"], minlength=len(coords))
    assert fan[octave == 1].max() > 4, "no anchor with more rows than one workgroup takes"
    # ... and the oracle's descriptors, octave by octave
    for o, (gx, gy, gz) in ((1, (gx1, gy1, gz1)), (0, (gx0, gy0, gz0))):
        rows = np.nonzero(octave[have["anchor"]] == o)[0]
        ref = O.describe(gx, gy, gz, o, coords[have["anchor"][rows]], have["R"][rows], E16.sphere_eqsp)
        np.testing.assert_array_equal(have["dsc"][rows], ref, err_msg="octave %d" % o)
    # a set rebuilt in place with the same anchors (the unchanged-anchors path keeps the sorting), then with other anchors
    job2 = ([s0, s1], coords[::-1].copy(), octave[::-1].copy(), subv[::-1].copy(), np.arange(len(coords), dtype=np.int32))
    lib.set_option("dsc_ball", 1)
    try:
        lib.set_build(*job, into=ball)
        np.testing.assert_array_equal(ball.download()["dsc"], want["dsc"])
        lib.set_build(*job2, into=ball)
    finally:
        lib.set_option("dsc_ball", 0)
    lib.set_build(*job2, into=rowwise)
    np.testing.assert_array_equal(ball.download()["dsc"], rowwise.download()["dsc"])
    ball.close()
    rowwise.close()


@pytest.mark.gpu
def test_partitioned_match_blocks_beside_an_open_bracket_equal_the_unsharded_topk(lib):
    """dist.PartitionedMatch on the device (12 subunits on 8 ranks: BASELINE configs[4]): every rank opens its asynchronous bracket of
    whole subunits (mad_match_topk_many_begin) and, with that bracket OPEN, runs its block of a leftover subunit through the
    synchronous mad_match_shard_pairs / _topk on lane 0 of the same context -- the same scratch slots, pinned words and status
    block a lane-0 match of the bracket uses.  All eight ranks of the plan are played in turn on this GPU, the group
    collectives replaced by stand-ins that carry the REAL flags and per-block lists from rank to rank (three rounds: flags,
    blocks, merge).  Every leftover subunit's merged rows must be the rows of mad_match_topk on the whole map set, and every whole
    subunit's rows must come out of the bracket untouched by the shard calls in between."""
    from mad_amd import dist as mdist
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    shape = (56, 60, 64)
    slot = lib.new_slot()
    lib.upload_field(slot, synth.gradient_field(synth.blob_volume(shape, n_blobs=60, seed=9, sigma=(1.5, 3.5))))
    rng = np.random.default_rng(5)

    def a_set(n, seed):
        coords = synth.interior_anchors(shape, n, 12, seed)
        return lib.set_build([-1, slot], coords, np.ones(n, np.int32), coords.astype(np.float64) * 1.5 + rng.normal(scale=0.2, size=(n, 3)), np.arange(n))

    n_items, world, cc, dist_, k = 12, 8, 0.45, 4.0, 30
    lo = a_set(150, 250)
    his = [a_set(40 + 5 * (i % 4), 300 + i) for i in range(n_items)]      # lanes 0..7 hold the first eight, lanes 0..3 the leftover four as well
    want = []
    for hi in his:
        top, idx, st = lib.match_topk(hi, lo, cc, dist_, k)
        assert st["n_pairs"] > 50
        want.append(top)
    units, groups = mdist.plan_partition(n_items, world)
    assert [len(g) for g in groups] == [2, 2, 2, 2] and all(len(u) == 2 for u in units)
    flags, blocks = {}, {}      # what the group collectives would carry: per leftover item, per part

    def play(rank, stage):
        """One rank's step; stage 0 records its flags, 1 its block's list (under the group's true flags), 2 merges the group's lists."""
        pm = mdist.PartitionedMatch(n_items, rank, world)
        item, part = pm.blocks[0][1], pm.blocks[0][2]

        def reduce_flags(f):
            f = np.asarray(f, dtype=np.uint8).copy()
            if stage == 0:
                flags.setdefault(item, {})[part] = f
                return f
            return np.bitwise_or.reduce(np.stack([flags[item][p] for p in sorted(flags[item])]), axis=0)

        def gather(parts):
            if stage == 1:
                blocks.setdefault(item, {})[part] = [np.array(x, copy=True) for x in parts]
            if stage < 2:
                return [parts]
            return [blocks[item][p] for p in sorted(blocks[item])]

        pm.stand_ins = (reduce_flags, gather)
        mine = [his[i] for i in pm.items]
        state = pm.begin(lib, mine, lo, cc, dist_, k)      # bracket open while the shard calls run
        corr, tops, stats = pm.finish(lib, state)
        return pm, tops

    for stage in (0, 1, 2):
        for rank in range(world):
            pm, tops = play(rank, stage)
            np.testing.assert_array_equal(tops[0], want[pm.items[0]], err_msg="whole subunit %d of rank %d, round %d" % (pm.items[0], rank, stage))
            if stage == 2 and mdist.owner_of(pm.blocks[0][1], world) == rank:      # the rank that reports the leftover subunit
                np.testing.assert_array_equal(tops[1], want[pm.blocks[0][1]], err_msg="leftover subunit %d merged on rank %d" % (pm.blocks[0][1], rank))
    assert sorted(blocks) == [8, 9, 10, 11] and all(sorted(b) == [0, 1] for b in blocks.values())
    for s in his + [lo]:
        s.close()


@pytest.mark.gpu
def test_two_phase_bounds_pass_selects_the_same_pairs_run_after_run(lib, fields):
    """k_pose_bounds in two launches (mad_set_option "pose_split" = 1): phase 2 abandons pairs against the threshold T_stop that the
    LAST workgroup of phase 1 leaves behind (an acq_rel ticket at agent scope; mad_match.hip).  With the lanes overlapped -- other
    matches running beside it, workgroups starting late -- the number of pairs that reach the exact search must be the same in every
    run (it was not while phase 2 read the histogram its own workgroups were still adding to), and the top-k with it."""
    f = fields[1]
    rng = np.random.default_rng(3)
    sets = []
    for n, seed in ((160, 41), (70, 42), (64, 43), (58, 44)):
        coords = synth.interior_anchors(f["shape"], n, 10, seed)
        sets.append(lib.set_build([-1, f["slot"]], coords, np.ones(n, np.int32), coords * 1.5 + rng.normal(scale=0.2, size=(n, 3)), np.arange(n)))
    lo, his = sets[0], sets[1:]
    cc, dist_, k = 0.3, 4.0, 20
    try:
        lib.set_option("pose_split", 1)
        lib.set_option("pose_split_min", 64)      # a first phase of a few pairs only: the second has something to abandon
        seen = []
        for it in range(20):
            out = lib.match_topk_many(his, lo, cc, dist_, k)      # three matches side by side on their lanes
            top, idx, st = lib.match_topk(his[0], lo, cc, dist_, k)
            seen.append((int(lib.last_pose_selected()), st["n_pairs"], top.tobytes(), tuple(o[0].tobytes() for o in out)))
            assert np.array_equal(out[0][0], top)
        assert seen[0][1] > 500 and 0 < seen[0][0] < seen[0][1]
        assert all(s == seen[0] for s in seen), sorted({s[0] for s in seen})
    finally:
        lib.set_option("pose_split", -1)
        lib.set_option("pose_split_min", 4096)
    for s in sets:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 3])
def test_sharded_match_without_host_round_trips_equals_unsharded(lib, parts):
    """mad_match_shard_begin / mad_match_shard_score (round 4): the two stages of a sharded match enqueued on the subunit's lane, the
    flags and the shard's record in device tensors (here torch's: what dist.ShardedMatchAsync hands over), the OR of the flags taken
    on the device between the two calls.  Merged over the shards the k best are those of mad_match_topk on the whole map set --
    rows, counts, order; a wrong row count of the map set and a pair capacity that is too small come back as flags, not as rows."""
    import torch
    from mad_amd import dist as mdist
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    shape = (56, 60, 64)
    slot = lib.new_slot()
    lib.upload_field(slot, synth.gradient_field(synth.blob_volume(shape, n_blobs=60, seed=9, sigma=(1.5, 3.5))))
    rng = np.random.default_rng(parts)
    sets = []
    for n in (150, 60):
        coords = synth.interior_anchors(shape, n, 12, 100 + n)
        sets.append(lib.set_build([-1, slot], coords, np.ones(n, np.int32), coords.astype(np.float64) * 1.5 + rng.normal(scale=0.2, size=(n, 3)), np.arange(n)))
    lo, hi = sets
    cc, dist_, k = 0.45, 4.0, 40
    top, idx, st = lib.match_topk(hi, lo, cc, dist_, k)
    assert st["n_pairs"] > 200 and len(top) == k
    ph, pl, _, cnt = lib.match_fetch(st["n_pairs"])
    n_lo = lo.size()[0]
    ref_rank = ph[idx].astype(np.int64) * n_lo + pl[idx]
    rec = lib.match_shard_record_doubles(k)
    # (every tensor is allocated up front, on torch's own stream, and the device is idle before the library writes into them)
    flags = [torch.zeros(hi.n_anchors + lo.n_anchors, dtype=torch.uint8, device="cuda") for _ in range(parts)]
    flags_all, own = torch.zeros_like(flags[0]), torch.zeros_like(flags[0])
    records = torch.zeros(parts, rec, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for p in range(parts):      # Exchange 1 needs every shard's flags ...
        b, e = mdist.lo_row_block(n_lo, p, parts)
        lib.match_shard_begin(hi, lo, b, e, n_lo, cc, flags[p].data_ptr())
    lib.synchronize()
    for f in flags:
        torch.maximum(flags_all, f, out=flags_all)
    torch.cuda.synchronize()
    for p in range(parts):      # ... then every shard again (one lane: the next shard's pairs replace this one's), scored under the OR
        b, e = mdist.lo_row_block(n_lo, p, parts)
        lib.match_shard_begin(hi, lo, b, e, n_lo, cc, own.data_ptr())
        lib.match_shard_score(hi, lo, flags_all.data_ptr(), dist_, k, records[p].data_ptr())
        lib.synchronize()
        assert torch.equal(own, flags[p])
    out = records.cpu().numpy()
    assert np.all(out[:, 1] == 0) and int(out[:, 3].sum()) == st["n_pairs"] and np.all(out[:, 2] == st["l_hi"])
    rows, cn, pr = [], [], []
    for r in out:
        m = int(r[0])
        rows.append(r[4:4 + m * 23].reshape(m, 23)); cn.append(r[4 + k * 23:4 + k * 23 + m].astype(np.int64)); pr.append(r[4 + k * 24:4 + k * 24 + m].astype(np.int64))
    mrows, mcnt, mrank = mdist.merge_topk(rows, cn, pr, k)
    np.testing.assert_array_equal(mrank, ref_rank)
    np.testing.assert_array_equal(mcnt, cnt[idx])
    np.testing.assert_allclose(mrows, top, rtol=0, atol=1e-12)
    # the object mad_amd.dist uses, alone on its GPU (no collective): one shard = the whole pair grid
    if parts == 1:
        h = mdist.ShardedMatchAsync(lib, hi, lo, cc, dist_, k, 0, 1, n_lo, local=True)
        r2, c2, p2 = h.finish()
        np.testing.assert_array_equal(p2, ref_rank)
        np.testing.assert_allclose(r2, top, rtol=0, atol=1e-12)
        # a map set with another row count than the blocks were cut from: refused by the device, reported, nothing returned
        bad = mdist.ShardedMatchAsync(lib, hi, lo, cc, dist_, k, 0, 1, n_lo - 1, local=True)
        assert bad.finish() is None
    for s in sets:
        s.close()
