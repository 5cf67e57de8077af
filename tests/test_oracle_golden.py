"""The CPU oracle (oracle/mad_oracle.c) against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then compare
the HIP path with the oracle and with the same fixtures.  Runs without a GPU."""
import os

import numpy as np
import pytest

from mad_amd import synth
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(G, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def g1():
    return load("g1_eqsp_math.npz")


def test_belts_and_matrices(g1):
    for n in (16, 112):
        bf = O.belt_first(g1["bounds_%d" % n])
        sizes = np.bincount(np.unique(bf, return_inverse=True)[1])
        np.testing.assert_array_equal(sizes, g1["belt_sizes_%d" % n])


@pytest.mark.parametrize("octave", [1, 0])
def test_orient_against_reference(g1, octave):
    g = load("g23_orient_describe.npz")
    grad = synth.gradient_field(g["vol_%d" % octave])
    got = O.orient(grad[..., 0], grad[..., 1], grad[..., 2], octave, g["coords_%d" % octave], g1["bounds_112"], g1["centers_112"])
    np.testing.assert_array_equal(got["anchor"], g["row_anchor_%d" % octave])
    np.testing.assert_array_equal(got["main"], g["row_main_%d" % octave])
    np.testing.assert_array_equal(got["sec"], g["row_sec_%d" % octave])
    np.testing.assert_array_equal(got["counts"], g["row_count_%d" % octave])
    np.testing.assert_allclose(got["R"], g["row_R_%d" % octave], rtol=0, atol=1e-14)
    for k in range(len(got["main"])):
        np.testing.assert_allclose(O.to_dom_mat(g1["centers_112"], int(got["main"][k])), g["row_dom_%d" % octave][k], atol=1e-14)
        np.testing.assert_allclose(O.adj_sec_mat(g1["bounds_112"], g1["centers_112"], int(got["sec"][k])), g["row_adj_%d" % octave][k], atol=1e-14)


def test_orient_rejects_counted():
    g = load("g23_orient_describe.npz")
    g1 = load("g1_eqsp_math.npz")
    tot = 0
    for octave in (1, 0):
        grad = synth.gradient_field(g["vol_%d" % octave])
        tot += O.orient(grad[..., 0], grad[..., 1], grad[..., 2], octave, g["coords_%d" % octave], g1["bounds_112"], g1["centers_112"])["n_reject"]
    assert tot == int(g["n_reject"]) and tot >= 2


@pytest.mark.parametrize("octave", [1, 0])
def test_describe_against_reference(g1, octave):
    g = load("g23_orient_describe.npz")
    grad = synth.gradient_field(g["vol_%d" % octave])
    got = O.describe(grad[..., 0], grad[..., 1], grad[..., 2], octave, g["dsc_coords_%d" % octave], g["dsc_R_%d" % octave], g1["bounds_16"])
    ref = g["dsc_%d" % octave]
    assert ref[-4].sum() == 0 and ref[-3].sum() == 0 and ref[-1].sum() > 0      # out-of-grid rows are zero, identity rows are not
    np.testing.assert_array_equal(got, ref)


def _meta(g, p):
    return np.stack([g[p + "index"], g[p + "oct"], g[p + "main"]], 1).astype(np.int32)


def test_match_against_reference():
    g = load("g4_match.npz")
    ref = g["results"]
    ph, pl, ps, _ = O.correlate(g["hi_dsc"], g["lo_dsc"], float(g["cc"]))
    assert len(ph) == len(ref) > 500
    np.testing.assert_allclose(ps, ref[:, 0], rtol=1e-12)
    # clouds as MaD.py:427-428
    hi_cloud = np.unique(g["hi_subv"][np.unique(ph)], axis=0)
    lo_cloud = np.unique(g["lo_subv"][np.unique(pl)], axis=0)
    np.testing.assert_array_equal(hi_cloud, g["hi_cloud"])
    np.testing.assert_array_equal(lo_cloud, g["lo_cloud"])
    res, cnt = O.pose_score(ph, pl, ps, g["hi_subv"], g["hi_R"], _meta(g, "hi_"), g["lo_subv"], g["lo_R"], _meta(g, "lo_"), hi_cloud, lo_cloud, 4.0)
    np.testing.assert_array_equal(res[:, 2:8], ref[:, 2:8])
    np.testing.assert_array_equal(res[:, 1], ref[:, 1])                  # repeatability: an integer count / l
    np.testing.assert_allclose(res[:, 8:14], ref[:, 8:14], rtol=0, atol=0)
    np.testing.assert_allclose(res[:, 14:], ref[:, 14:], rtol=0, atol=1e-13)
    # the sort MaD._filter_dsc_pairs applies (MaD.py:480)
    order = O.topk(cnt, 120)
    py = sorted(range(len(ref)), key=lambda i: ref[i][1], reverse=True)[:120]
    np.testing.assert_array_equal(order, py)


def test_refine_against_reference():
    g = load("g6_refine.npz")
    origin, vs = g["map_origin"], float(g["map_vs"])
    for tag in ("a", "b"):
        for n in (1, 2, 3, 4, 5, 8, 500):
            ref = g["final_%s_%d" % (tag, n)]
            rmsd, conv, step = g["ret_%s_%d" % (tag, n)]
            got, gconv, glast, _ = O.refine(g["map_grid"], origin, vs, g["start_" + tag], n_steps=n, max_step=1.0, min_step=0.1)
            assert (gconv, glast) == (bool(conv), int(step)), (tag, n)
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-8 if n <= 8 else 1e-6)
            ca = g["ca_idx"]
            assert abs(np.sqrt(np.sum((got[ca] - g["start_" + tag][ca]) ** 2) / len(ca)) - rmsd) < 1e-6


def test_density_against_reference():
    g = load("g7_density_ccc.npz")
    m = synth.masses([str(e) for e in g["elements"]])
    vs, res = float(g["vs"]), float(g["res"])
    sp, dims, mn = O.splat(g["atoms"], m, vs)
    ref_sp = g["splat"]
    assert tuple(dims) == ref_sp.shape
    np.testing.assert_allclose(mn, g["splat_min"], atol=0)
    np.testing.assert_allclose(np.reshape(sp, ref_sp.shape, order="F"), ref_sp, rtol=0, atol=1e-15)
    dens, x0, y0, z0 = O.structure_to_density(g["atoms"], m, res, vs)
    np.testing.assert_allclose([x0, y0, z0], g["density_origin"], atol=0)
    np.testing.assert_allclose(dens, g["density"], rtol=0, atol=1.5e-7)
    dens2, _, _, _ = O.structure_to_density(g["atoms"], m, 6.0, 1.2, isovalue=0.05)
    np.testing.assert_allclose(dens2, g["density_iso"], rtol=0, atol=1.5e-7)
    assert np.all((dens2 == 0) == (g["density_iso"] == 0))


def test_ccc_against_reference():
    g = load("g7_density_ccc.npz")
    for sh, ref in zip(g["ccc_shifts"], g["ccc"]):
        a, b = g["map_grid"].copy(), g["density"].copy()
        got = O.ccc(a, g["map_origin"], b, g["density_origin"] + sh, float(g["map_vs"]))
        if np.isnan(ref):
            assert np.isnan(got)
        else:
            assert abs(got - ref) <= 1e-5 * max(abs(ref), 1e-3), (sh, got, ref)
    assert g["ccc"][0] > 0.5 and g["ccc"][-1] == 0


def test_g10_fields_made_of_zone_bounds():
    """The oracle on the reference's own outputs for a gradient field whose every voxel points at an edge of the 16-zone
    table (tests/golden/make_golden.py::make_g10): Orientator rows (bins and 112-zone histograms) and Descriptor rows,
    identical.  This is the fixture that pins the sliver where two zones overlap (a voxel there counts in both zones of the
    Orientator) and the bit-exact to_dom matrices."""
    from mad_amd.eqsp import EQSP_Sphere
    from oracle import oracle as O
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_bounds.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    f = g["field"]
    rows = O.orient(f[0], f[1], f[2], 1, g["coords"], e112.sphere_eqsp, e112.p_centers_eqsp)
    assert len(g["row_anchor"]) > 40
    np.testing.assert_array_equal(rows["anchor"], g["row_anchor"])
    np.testing.assert_array_equal(rows["main"], g["row_main"])
    np.testing.assert_array_equal(rows["sec"], g["row_sec"])
    np.testing.assert_array_equal(rows["counts"], g["row_count"])
    dsc = O.describe(f[0], f[1], f[2], 1, g["coords"], g["dsc_R"], e16.sphere_eqsp)
    np.testing.assert_array_equal(dsc, g["dsc"])


def test_g11_pose_count_at_the_distance_threshold():
    """MaD._match_dsc of the reference on a hi cloud that sits on the 4 A decision surface (exactly 4, one ulp either side,
    +-1e-12): the oracle's counts and result rows."""
    from oracle import oracle as O
    g = load("g11_pose_threshold.npz")
    M, N = len(g["hi_p"]), len(g["lo_p"])
    ph, pl = np.divmod(np.arange(M * N), N)
    eye = np.tile(np.identity(3), (max(M, N), 1, 1))
    meta_h = np.stack([np.arange(M), np.ones(M), np.full(M, 3)], 1).astype(np.int32)
    meta_l = np.stack([np.arange(N), np.ones(N), np.full(N, 3)], 1).astype(np.int32)
    ref = g["results"]
    assert len(ref) == M * N and len(np.unique(ref[:, 1])) > 8
    want_cnt = np.rint(ref[:, 1] * len(g["hi_cloud"]) / 100.0).astype(np.int32)
    res, cnt = O.pose_score(ph.astype(np.int32), pl.astype(np.int32), ref[:, 0].copy(), g["hi_p"], eye[:M], meta_h, g["lo_p"], eye[:N], meta_l,
                            g["hi_cloud"], g["lo_cloud"], dist=4.0)
    np.testing.assert_array_equal(cnt, want_cnt)
    np.testing.assert_allclose(res, ref, rtol=0, atol=1e-12)


def test_g17_stage_options_of_the_constructors():
    """Orientator(gw_sig=...) and Descriptor(dsc_size=27 | 8 | 1) -- options MaD.run never selects (tests/golden/make_golden.py::
    make_g17): the oracle on the reference's own outputs, both octaves.  With a window the reference stores each zone's float64
    weight sum into an int32 array, i.e. truncates it (DensityFeature.py:50): rows, bins and quantised histograms identical."""
    from mad_amd.eqsp import EQSP_Sphere
    from oracle import oracle as O
    g = load("g17_options.npz")
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    for octave in (1, 0):
        f, coords = g["o%d_field" % octave], g["o%d_coords" % octave]
        for gw in (2.0, 4.5):
            k = "o%d_gw%g_" % (octave, gw)
            rows = O.orient(f[0], f[1], f[2], octave, coords, e112.sphere_eqsp, e112.p_centers_eqsp, gw_sig=gw)
            assert len(g[k + "anchor"]) > 60
            np.testing.assert_array_equal(rows["anchor"], g[k + "anchor"])
            np.testing.assert_array_equal(rows["main"], g[k + "main"])
            np.testing.assert_array_equal(rows["sec"], g[k + "sec"])
            np.testing.assert_array_equal(rows["counts"], g[k + "count"])
            np.testing.assert_allclose(rows["R"], g[k + "R"], rtol=0, atol=1e-14)
        for size in (27, 8, 1):
            want = g["o%d_dsc%d" % (octave, size)]
            assert want.shape[1] == size * 16 and want.sum() > 30000
            np.testing.assert_array_equal(O.describe(f[0], f[1], f[2], octave, coords, g["o%d_dsc_R" % octave], e16.sphere_eqsp, dsc_size=size), want)


def test_g18_the_other_eqsp_sizes():
    """Orientator(eqsp_size=16) -- the "coarse eqsp" of BASELINE configs[0] -- and Descriptor(subeqsp_size=112) (the reference ships
    both tables, eqsp.py:16; tests/golden/make_golden.py::make_g18): the oracle on the reference's own outputs, both octaves, a
    border reject included."""
    from mad_amd import synth
    from mad_amd.eqsp import EQSP_Sphere
    from oracle import oracle as O
    g = load("g18_eqsp_sizes.npz")
    e112, e16 = EQSP_Sphere(112), EQSP_Sphere(16)
    for octave in (1, 0):
        f = np.ascontiguousarray(np.moveaxis(synth.gradient_field(g["o%d_vol" % octave]), -1, 0))
        coords = g["o%d_coords" % octave]
        k = "o%d_ori16_" % octave
        rows = O.orient(f[0], f[1], f[2], octave, coords, e16.sphere_eqsp, e16.p_centers_eqsp)
        assert len(g[k + "anchor"]) > 60 and rows["counts"].shape[1] == 16
        assert rows["n_reject"] == int(g[k + "reject"]) == 1
        np.testing.assert_array_equal(rows["anchor"], g[k + "anchor"])
        np.testing.assert_array_equal(rows["main"], g[k + "main"])
        np.testing.assert_array_equal(rows["sec"], g[k + "sec"])
        np.testing.assert_array_equal(rows["counts"], g[k + "count"])
        np.testing.assert_allclose(rows["R"], g[k + "R"], rtol=0, atol=1e-14)
        want = g["o%d_dsc112" % octave]
        assert want.shape[1] == 64 * 112 and want.sum() > 30000
        np.testing.assert_array_equal(O.describe(f[0], f[1], f[2], octave, coords[:len(want)], g["o%d_dsc_R" % octave], e112.sphere_eqsp), want)
