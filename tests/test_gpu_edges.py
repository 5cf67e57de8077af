"""Edge cases of the C-ABI: empty and degenerate inputs, thresholds that select nothing or everything, k beyond the
list, ties, anchors on the border, sample cubes that leave the grid, zero descriptors."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mad_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(lib):
    from mad_amd.eqsp import EQSP_Sphere
    from mad_amd.orient_tables import orientation_matrices
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    dom, adj = orientation_matrices(e112)
    lib.set_eqsp(0, e112.sphere_eqsp, dom, adj)
    lib.set_eqsp(1, e16.sphere_eqsp)
    shape = (40, 42, 44)
    vol = synth.blob_volume(shape, n_blobs=40, seed=3, sigma=(1.5, 3.5))
    g = synth.gradient_field(vol)
    slot = lib.new_slot()
    lib.upload_field(slot, g)
    yield dict(lib=lib, slot=slot, shape=shape, g=g, e112=e112, e16=e16)
    lib.free_field(slot)


def test_orient_empty_and_all_rejected(ctx):
    lib, slot, shape = ctx["lib"], ctx["slot"], ctx["shape"]
    out = lib.orient(slot, 1, np.zeros((0, 3), np.int32))
    assert len(out["anchor"]) == 0 and out["n_reject"] == 0
    # every anchor closer than r to a face: the reference rejects them all (Orientator.py:131-135)
    coords = np.array([[3, 20, 20], [20, 3, 20], [20, 20, 40], [36, 20, 20]], np.int32)
    out = lib.orient(slot, 1, coords)
    assert len(out["anchor"]) == 0 and out["n_reject"] == len(coords)
    ref = O.orient(ctx["g"][..., 0], ctx["g"][..., 1], ctx["g"][..., 2], 1, coords, ctx["e112"].sphere_eqsp, ctx["e112"].p_centers_eqsp)
    assert len(ref["anchor"]) == 0 and ref["n_reject"] == len(coords)


def test_describe_empty_and_out_of_grid(ctx):
    lib, slot, shape = ctx["lib"], ctx["slot"], ctx["shape"]
    assert lib.describe(slot, 1, np.zeros((0, 3), np.int32), np.zeros((0, 3, 3))).shape == (0, 1024)
    rng = np.random.default_rng(1)
    R = np.stack([synth.random_rotation(rng) for _ in range(3)])
    coords = np.array([[2, 20, 20], [20, 21, 22], [38, 40, 42]], np.int32)      # first and last: the cube leaves the grid
    d = lib.describe(slot, 1, coords, R)
    assert not d[0].any() and not d[2].any() and d[1].sum() > 3000      # Descriptor.py:140-149: zero descriptor, no error
    np.testing.assert_array_equal(d, O.describe(ctx["g"][..., 0], ctx["g"][..., 1], ctx["g"][..., 2], 1, coords, R, ctx["e16"].sphere_eqsp))


def test_correlate_thresholds_and_empty_sides(ctx):
    lib = ctx["lib"]
    rng = np.random.default_rng(2)
    hi = rng.integers(0, 9, (37, 1024)).astype(np.int16)
    lo = rng.integers(0, 9, (53, 1024)).astype(np.int16)
    lo[5] = 0      # a zero row stays un-normalised (MaD.py:416): its score is 0, never NaN
    for a, b in ((hi[:0], lo), (hi, lo[:0])):
        ph, pl, ps = lib.correlate(a, b, 0.5)
        assert len(ph) == len(pl) == len(ps) == 0
    ph, pl, ps = lib.correlate(hi, lo, 1.5)      # nothing exceeds a score of 1.5
    assert len(ph) == 0
    ph, pl, ps = lib.correlate(hi, lo, -1.0)     # everything does, in row-major order
    assert len(ph) == 37 * 53
    np.testing.assert_array_equal(ph, np.repeat(np.arange(37), 53))
    np.testing.assert_array_equal(pl, np.tile(np.arange(53), 37))
    assert np.all(ps[pl == 5] == 0.0) and np.isfinite(ps).all()
    rph, rpl, rps, _ = O.correlate(hi, lo, -1.0)
    np.testing.assert_allclose(ps, rps, rtol=1e-12, atol=0)


def test_topk_degenerate(ctx):
    lib = ctx["lib"]
    assert len(lib.topk(np.zeros(0, np.int32), 5)) == 0
    c = np.array([3, 3, 3, 3], np.int32)
    np.testing.assert_array_equal(lib.topk(c, 10), [0, 1, 2, 3])      # k beyond the list, all ties: input order
    np.testing.assert_array_equal(lib.topk(np.array([0, 5, 0, 5, 1], np.int32), 3), [1, 3, 4])
    np.testing.assert_array_equal(lib.topk(np.array([7], np.int32), 1), [0])


def test_sets_and_matches_with_nothing_in_them(ctx):
    lib, slot, shape = ctx["lib"], ctx["slot"], ctx["shape"]
    n = 30
    coords = synth.interior_anchors(shape, n, 12, 5)
    subv = coords.astype(np.float64) * 1.5
    full = lib.set_build([-1, slot], coords, np.ones(n, np.int32), subv, np.arange(n))
    empty = lib.set_build([-1, slot], np.zeros((0, 3), np.int32), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros(0, np.int32))
    assert empty.size() == (0, 0) and full.size()[0] > 0
    for hi, lo in ((empty, full), (full, empty), (empty, empty)):
        top, idx, st = lib.match_topk(hi, lo, 0.5, 4.0, 10)
        assert len(top) == 0 and len(idx) == 0 and st["n_pairs"] == 0
    top, idx, st = lib.match_topk(full, full, 1.5, 4.0, 10)      # no pair above the threshold
    assert len(top) == 0 and st["n_pairs"] == 0
    assert lib.match_topk_many([], full, 0.5, 4.0, 10) == []
    res = lib.match_topk_many([empty, full, empty], full, 0.5, 4.0, 10)
    assert [len(r[0]) for r in res][0] == 0 and len(res[1][0]) == 10 and len(res[2][0]) == 0
    # the two-call form: same answer; a second bracket while one is open, or a finish without a begin, is refused
    from mad_amd._lib import MadBackendError
    h = lib.match_topk_many_begin([empty, full, empty], full, 0.5, 4.0, 10)
    other = lib.set_build([-1, slot], coords, np.ones(n, np.int32), subv, np.arange(n))      # building another set in between is allowed
    h2 = lib.match_topk_many_begin([other, other], full, 0.5, 4.0, 7)      # ... and so are a second bracket
    h3 = lib.match_topk_many_begin([full, empty], full, 0.5, 4.0, 4)       # and a third (four steps in flight)
    with pytest.raises(MadBackendError):
        lib.match_topk_many_begin([full], full, 0.5, 4.0, 10)             # a fourth is not
    with pytest.raises(MadBackendError):
        lib.match_topk_many_finish(h2)                                    # they finish in the order they began
    res2 = lib.match_topk_many_finish(h)
    for a, b in zip(res, res2):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    h4 = lib.match_topk_many_begin([full], full, 0.5, 4.0, 3)             # the ring of slots wraps around
    res3 = lib.match_topk_many_finish(h2)
    np.testing.assert_array_equal(res3[0][0], res[1][0][:7])
    np.testing.assert_array_equal(res3[1][0], res[1][0][:7])
    res4 = lib.match_topk_many_finish(h3)
    np.testing.assert_array_equal(res4[0][0], res[1][0][:4])
    assert len(res4[1][0]) == 0
    res5 = lib.match_topk_many_finish(h4)
    np.testing.assert_array_equal(res5[0][0], res[1][0][:3])
    with pytest.raises(MadBackendError):
        lib.match_topk_many_finish(h)
    np.testing.assert_array_equal(other.download()["dsc"], full.download()["dsc"])
    other.close()
    # a set against itself: every row pairs with itself at score 1 and rebuilds the identity pose
    top, idx, st = lib.match_topk(full, full, 0.999999, 4.0, 5)
    assert len(top) == 5 and np.allclose(top[:, 0], 1.0) and np.allclose(top[:, 1], 100.0)
    np.testing.assert_allclose(top[:, 14:23].reshape(-1, 3, 3), np.broadcast_to(np.eye(3), (5, 3, 3)), atol=1e-12)
    for s in (full, empty):
        s.close()


def test_refine_density_ccc_degenerate(ctx):
    lib = ctx["lib"]
    atoms, names, elems = synth.random_globule(200, 8.0, seed=4)
    m = synth.masses(elems)
    grid, x0, y0, z0 = lib.structure_to_density(atoms, m, 8.0, 2.0)
    lib.upload_density(grid, (x0, y0, z0), 2.0)
    same, conv, last = lib.refine(atoms, n_steps=0)
    np.testing.assert_array_equal(same, atoms)      # no step taken, nothing moved
    one, _, _, _ = lib.structure_to_density(atoms[:1], m[:1], 8.0, 2.0)
    assert one.max() == pytest.approx(1.0) and one.min() >= 0.0
    far = lib.ccc(grid.copy(), np.array([x0, y0, z0]), grid.copy(), np.array([x0 + 1e4, y0, z0]), 2.0)
    assert far == 0.0      # no overlap (Dmap.py:232-234)
    assert lib.ccc(grid.copy(), np.array([x0, y0, z0]), grid.copy(), np.array([x0, y0, z0]), 2.0) == pytest.approx(1.0, abs=1e-12)
    assert lib.density_ccc(np.zeros((0, len(atoms), 3)), m, 8.0).shape == (0,)


def test_scale_space_of_a_tiny_grid(ctx):
    from mad_amd._lib import DeviceSpace
    from oracle import scale_space as OS
    lib = ctx["lib"]
    rng = np.random.default_rng(5)
    grid = rng.random((5, 4, 6)).astype(np.float32)
    ref = OS.build_volumes(grid, pad=9, oct_mode="both", sig_init=2, sig_presmooth=1)
    sp = DeviceSpace(lib).build(grid, pad=9)
    np.testing.assert_array_equal(sp.download(1, DeviceSpace.LOG), ref["map_space"][1])
    np.testing.assert_allclose(sp.download(0, DeviceSpace.LOG), ref["map_space"][0], rtol=0, atol=1e-6)
    coords, vals = sp.peaks(1, threshold=5e-2, border=12)
    np.testing.assert_array_equal(coords, OS.peak_local_max(ref["map_space"][1], 12, 5e-2))
    sp.close()
