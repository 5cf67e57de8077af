"""GPU parity against the REFERENCE's own outputs (tests/golden/*.npz), through the C-ABI
and through the drop-in Python classes."""
import os
import types

import numpy as np
import pytest

from mad_amd import synth
from mad_amd.DensityFeature import DensityFeature

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(G, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def default_lib(lib):
    """Route the drop-in classes (which use the process-wide context) to the session context."""
    from mad_amd import _lib
    old = _lib._default
    _lib._default = lib
    lib._eq_loaded = {}
    yield lib
    _lib._default = old


class _FakeSpace(object):
    """The attributes of MapSpace the stages read, around two synthetic fields."""

    def __init__(self, g):
        self.grad_list = [synth.gradient_field(g["vol_0"]), synth.gradient_field(g["vol_1"])]
        self.oct_mode = "both"
        self._slots = None
        self.name = "golden"

    def device_slots(self, lib):
        if self._slots is None:
            slots = []
            for g in self.grad_list:
                slots.append(lib.new_slot())
                lib.upload_field(slots[-1], g)
            self._slots = slots
        return self._slots


def _anchors(coords, octave):
    out = []
    for i, c in enumerate(coords):
        df = DensityFeature()
        df.set_detector_info(i, octave, [int(c[0]), int(c[1]), int(c[2])], np.array(c, float), np.array(c, float) + 0.25, 1.0)
        out.append(df)
    return out


def test_orientator_and_descriptor_classes(default_lib):
    from mad_amd.Descriptor import Descriptor
    from mad_amd.Orientator import Orientator
    g = load("g23_orient_describe.npz")
    ms = _FakeSpace(g)
    ori, dsc = Orientator(), Descriptor()
    for octave in (1, 0):
        rows = ori.assign_orientations(ms, _anchors(g["coords_%d" % octave], octave))
        np.testing.assert_array_equal([r.index for r in rows], g["row_anchor_%d" % octave])
        np.testing.assert_array_equal([r.main_bin for r in rows], g["row_main_%d" % octave])
        np.testing.assert_array_equal([r.sec_bin for r in rows], g["row_sec_%d" % octave])
        np.testing.assert_array_equal([r.ar_count for r in rows], g["row_count_%d" % octave])
        np.testing.assert_allclose([r.Rfinal for r in rows], g["row_R_%d" % octave], rtol=0, atol=1e-14)
        np.testing.assert_allclose([r.to_dom_mat for r in rows], g["row_dom_%d" % octave], rtol=0, atol=1e-15)
        np.testing.assert_allclose([r.adj_sec_mat for r in rows], g["row_adj_%d" % octave], rtol=0, atol=1e-15)
        # descriptors, including rows that leave the grid and identity rotations
        drows = []
        for c, R in zip(g["dsc_coords_%d" % octave], g["dsc_R_%d" % octave]):
            df = DensityFeature()
            df.set_detector_info(0, octave, [int(v) for v in c], np.zeros(3), np.zeros(3), 1.0)
            df.Rfinal = R
            drows.append(df)
        dsc.generate_descriptors(ms, drows)
        np.testing.assert_array_equal([r.lin_ar_subeqsp for r in drows], g["dsc_%d" % octave])
        assert drows[0].lin_ar_subeqsp.dtype == np.int16
    assert ori.step1_reject == int(g["n_reject"])


def _rows(g, p):
    out = []
    for i in range(len(g[p + "index"])):
        df = DensityFeature()
        df.set_from_file_dsc(int(g[p + "index"][i]), int(g[p + "main"][i]), int(g[p + "sec"][i]), int(g[p + "oct"][i]), 112, 16,
                             g[p + "coords"][i], g[p + "subv"][i], g[p + "subv"][i], g[p + "R"][i], g[p + "dsc"][i])
        out.append(df)
    return out


def test_match_dsc_against_reference(default_lib):
    from mad_amd.MaD import MaD
    g = load("g4_match.npz")
    ref = g["results"]
    lo, hi = _rows(g, "lo_"), _rows(g, "hi_")
    m = MaD()
    results, lo_cloud, hi_cloud = m._match_dsc(lo, hi, cc_threshold=float(g["cc"]))
    results = np.array(results)
    assert results.shape == ref.shape
    np.testing.assert_array_equal(lo_cloud, g["lo_cloud"])
    np.testing.assert_array_equal(hi_cloud, g["hi_cloud"])
    np.testing.assert_allclose(results[:, 0], ref[:, 0], rtol=1e-12)
    np.testing.assert_array_equal(results[:, 1:14], ref[:, 1:14])
    np.testing.assert_allclose(results[:, 14:], ref[:, 14:], rtol=0, atol=1e-13)
    # device top-k == the reference's stable sort by repeatability (MaD.py:480), identity and order
    for k in (1, 17, 120, len(ref), len(ref) + 50):
        top, _, _ = m._match_dsc_topk(lo, hi, k, cc_threshold=float(g["cc"]))
        py = sorted(range(len(ref)), key=lambda i: ref[i][1], reverse=True)[:k]
        np.testing.assert_array_equal(top[:, 1:14], ref[py][:, 1:14])
        np.testing.assert_allclose(top[:, 0], ref[py][:, 0], rtol=1e-12)


def test_filter_refine_against_reference(default_lib, tmp_path, monkeypatch):
    import mad_amd.MaD as M
    from mad_amd.Dmap import Dmap
    g4, g5, g7, g8 = load("g4_match.npz"), load("g5_filter.npz"), load("g7_density_ccc.npz"), load("g8_solutions.npz")
    names = [synth.ATOM_CYCLE[i % 4][0] for i in range(len(g5["atoms"]))]
    pdbfile = str(tmp_path / "sub.pdb")
    synth.write_pdb(pdbfile, g5["atoms"], names, [str(e) for e in g7["elements"]])

    # the processed map exactly as the reference's Dmap held it (no text round trip)
    def fixture_map(_path):
        d = Dmap.__new__(Dmap)
        d.grid3d = g7["map_grid"].copy()
        d.xi, d.yi, d.zi = (float(v) for v in g7["map_origin"])
        d.xb, d.yb, d.zb = d.grid3d.shape
        d.voxsp = float(g7["map_vs"])
        d.map_name, d.name = "map.sit", "map"
        return d

    monkeypatch.setattr(M, "Dmap", fixture_map)
    m = M.MaD()
    m.processed_map, m.resolution, m.map_name = "map.sit", float(g7["res"]), "map"
    lo, hi = _rows(g4, "lo_"), _rows(g4, "hi_")
    top, lo_cloud, hi_cloud = m._match_dsc_topk(lo, hi, 120, cc_threshold=float(g4["cc"]))
    filt = m._filter_dsc_pairs(pdbfile, top, lo_cloud, hi_cloud, wthresh=4, n_samples=120, presorted=True)
    assert len(filt) == int(g5["n"])
    np.testing.assert_array_equal([f[4] for f in filt], g5["weight"])
    np.testing.assert_array_equal([f[5] for f in filt], g5["repeat"])
    np.testing.assert_allclose([f[7].coords for f in filt], g5["placed"], rtol=0, atol=1e-9)
    final = m._refine_filtered_solutions(pdbfile, filt, lo_cloud, hi_cloud)
    assert len(final) == int(g8["n"])
    np.testing.assert_array_equal([f[3] for f in final], g8["weight"])
    np.testing.assert_allclose([f[2] for f in final], g8["repeat"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([f[0].coords for f in final], g8["coords"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([f[4] for f in final], g8["ccc"], rtol=1e-5)
    np.testing.assert_allclose([f[6] for f in final], g8["score"], rtol=1e-5)
    # and both planted copies are found
    for t in g8["truth"]:
        assert min(np.sqrt(((f[0].coords - t) ** 2).sum(1).mean()) for f in final) < 1.0


def test_refine_against_reference(lib):
    g = load("g6_refine.npz")
    lib.upload_density(g["map_grid"], g["map_origin"], float(g["map_vs"]))
    for tag in ("a", "b"):
        for n in (1, 2, 3, 4, 5, 8, 500):
            ref = g["final_%s_%d" % (tag, n)]
            _, conv, step = g["ret_%s_%d" % (tag, n)]
            got, gconv, glast = lib.refine(g["start_" + tag], n_steps=n, max_step=1.0, min_step=0.1)
            assert (gconv, glast) == (bool(conv), int(step)), (tag, n)
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-8 if n <= 8 else 1e-6)


def test_density_ccc_against_reference(lib):
    g = load("g7_density_ccc.npz")
    m = synth.masses([str(e) for e in g["elements"]])
    dens, x0, y0, z0 = lib.structure_to_density(g["atoms"], m, float(g["res"]), float(g["vs"]))
    np.testing.assert_allclose([x0, y0, z0], g["density_origin"], atol=0)
    np.testing.assert_allclose(dens, g["density"], rtol=0, atol=2e-7)
    dens2, _, _, _ = lib.structure_to_density(g["atoms"], m, 6.0, 1.2, isovalue=0.05)
    np.testing.assert_allclose(dens2, g["density_iso"], rtol=0, atol=2e-7)
    for sh, ref in zip(g["ccc_shifts"], g["ccc"]):
        a, b = g["map_grid"].copy(), g["density"].copy()
        got = lib.ccc(a, g["map_origin"], b, g["density_origin"] + sh, float(g["map_vs"]))
        assert abs(got - ref) <= 1e-5 * max(abs(ref), 1e-3)      # north_star tolerance: 1e-5 relative


def test_density_ccc_batch_on_device(lib):
    """mad_density_ccc (density simulation + CCC of a batch of placed copies without leaving the device) against the
    reference's CCC values (atoms shifted by whole voxels = the reference's shifted origin) and against the
    stage-by-stage path."""
    g = load("g7_density_ccc.npz")
    vs = float(g["map_vs"])
    assert vs == float(g["vs"])
    m = synth.masses([str(e) for e in g["elements"]])
    shifts = g["ccc_shifts"]
    assert np.allclose(shifts[:3] / vs, np.round(shifts[:3] / vs)) and g["ccc"][3] == 0      # the last one leaves the map
    lib.upload_density(g["map_grid"], g["map_origin"], vs)
    cands = np.stack([g["atoms"] + sh for sh in shifts] + [g["atoms"] + np.array([0.37, -0.81, 1.13])])
    got = lib.density_ccc(cands, m, float(g["res"]))
    for v, ref in zip(got[:len(shifts)], g["ccc"]):
        assert abs(v - ref) <= 1e-5 * max(abs(ref), 1e-3)      # north_star tolerance: 1e-5 relative
    for c, v in zip(cands, got):
        dens, x0, y0, z0 = lib.structure_to_density(c, m, float(g["res"]), vs)
        ref = lib.ccc(g["map_grid"].copy(), g["map_origin"], dens, np.array([x0, y0, z0]), vs)
        assert abs(v - ref) <= 1e-12 * max(abs(ref), 1.0)


def test_pdb_and_dmap_classes(default_lib, tmp_path):
    from mad_amd.Dmap import Dmap
    from mad_amd.PDB import PDB
    g = load("g7_density_ccc.npz")
    names = [synth.ATOM_CYCLE[i % 4][0] for i in range(len(g["atoms"]))]
    pdbfile = str(tmp_path / "s.pdb")
    synth.write_pdb(pdbfile, g["atoms"], names, [str(e) for e in g["elements"]])
    p = PDB(pdbfile)
    assert p.n_atoms == len(g["atoms"]) and len(p.CA_idx) == (len(g["atoms"]) + 2) // 4
    np.testing.assert_array_equal(p.coords, g["atoms"])
    grid, x0, y0, z0 = p.structure_to_density(float(g["res"]), float(g["vs"]))
    np.testing.assert_allclose(grid, g["density"], rtol=0, atol=2e-7)
    from mad_amd import mapio
    mapfile = str(tmp_path / "m.mrc")
    mapio.write_mrc(mapfile, g["map_grid"], g["map_origin"], float(g["map_vs"]))
    d = Dmap(mapfile)
    assert d.grid3d.shape == g["map_grid"].shape and abs(d.voxsp - float(g["map_vs"])) < 1e-6
    ccc = d.get_CCC_with_grid(grid, x0, y0, z0)
    # Dmap truncates the MRC origin to integers exactly like the reference (Dmap.py:38): only sanity here
    assert 0.0 < ccc <= 1.0


def test_g10_fields_made_of_zone_bounds(lib):
    """The HIP path on the reference's outputs for the field of zone bounds (see tests/test_oracle_golden.py)."""
    g = load("g10_bounds.npz")
    slot = lib.new_slot()
    lib.upload_field(slot, g["field"])
    try:
        rows = lib.orient(slot, 1, g["coords"])
        np.testing.assert_array_equal(rows["anchor"], g["row_anchor"])
        np.testing.assert_array_equal(rows["main"], g["row_main"])
        np.testing.assert_array_equal(rows["sec"], g["row_sec"])
        np.testing.assert_array_equal(rows["counts"], g["row_count"])
        np.testing.assert_array_equal(lib.describe(slot, 1, g["coords"], g["dsc_R"]), g["dsc"])
    finally:
        lib.free_field(slot)


def test_g11_pose_count_at_the_distance_threshold(lib):
    """The HIP pose scoring (bitmap prefilter + exact search) on the reference's counts for a hi cloud on the 4 A decision surface."""
    g = load("g11_pose_threshold.npz")
    M, N = len(g["hi_p"]), len(g["lo_p"])
    ph, pl = np.divmod(np.arange(M * N), N)
    eye = np.tile(np.identity(3), (max(M, N), 1, 1))
    meta_h = np.stack([np.arange(M), np.ones(M), np.full(M, 3)], 1).astype(np.int32)
    meta_l = np.stack([np.arange(N), np.ones(N), np.full(N, 3)], 1).astype(np.int32)
    ref = g["results"]
    assert len(ref) == M * N and len(np.unique(ref[:, 1])) > 8
    want_cnt = np.rint(ref[:, 1] * len(g["hi_cloud"]) / 100.0).astype(np.int32)
    res, cnt = lib.pose_score(ph.astype(np.int32), pl.astype(np.int32), ref[:, 0].copy(), g["hi_p"], eye[:M], meta_h, g["lo_p"], eye[:N], meta_l,
                              g["hi_cloud"], g["lo_cloud"], dist=4.0)
    np.testing.assert_array_equal(cnt, want_cnt)
    np.testing.assert_allclose(res, ref, rtol=0, atol=1e-12)


def test_stage_options_of_the_constructors(default_lib):
    """Orientator(gw_sig=...), Orientator(magn_weighted=True) and Descriptor(dsc_size=27 | 8 | 1) through the drop-in classes
    against the reference's own outputs (g17): rows, bins, quantised 112-zone histograms and descriptors identical; and the
    shorter descriptor rows go through the correlation kernel while their counts fit its int8 operands (rows padded to its K step)."""
    from mad_amd.Descriptor import Descriptor
    from mad_amd.Orientator import Orientator
    g = load("g17_options.npz")
    lib = default_lib
    for octave in (1, 0):
        field = np.ascontiguousarray(np.moveaxis(g["o%d_field" % octave], 0, -1))
        ms = types.SimpleNamespace(grad_list=[field, field], oct_mode="both", name="g17")
        slot = lib.new_slot()
        lib.upload_field(slot, field)
        ms.device_slots = lambda lib_, s=slot: [s, s]
        coords = g["o%d_coords" % octave]
        for gw in (2.0, 4.5):
            k = "o%d_gw%g_" % (octave, gw)
            rows = Orientator(gw_sig=gw, magn_weighted=True).assign_orientations(ms, _anchors(coords, octave))
            np.testing.assert_array_equal([r.index for r in rows], g[k + "anchor"])
            np.testing.assert_array_equal([r.main_bin for r in rows], g[k + "main"])
            np.testing.assert_array_equal([r.sec_bin for r in rows], g[k + "sec"])
            np.testing.assert_array_equal([r.ar_count for r in rows], g[k + "count"])
            np.testing.assert_allclose([r.Rfinal for r in rows], g[k + "R"], rtol=0, atol=1e-14)
        plain = Orientator().assign_orientations(ms, _anchors(coords, octave))      # the window is off again afterwards
        assert len(plain) != len(g["o%d_gw2_anchor" % octave]) or [r.main_bin for r in plain] != list(g["o%d_gw2_main" % octave])
        for size in (27, 8, 1):
            drows = []
            for c, R in zip(coords, g["o%d_dsc_R" % octave]):
                df = DensityFeature()
                df.set_detector_info(0, octave, [int(v) for v in c], np.zeros(3), np.zeros(3), 1.0)
                df.Rfinal = R
                drows.append(df)
            Descriptor(dsc_size=size).generate_descriptors(ms, drows)
            got = np.array([r.lin_ar_subeqsp for r in drows])
            np.testing.assert_array_equal(got, g["o%d_dsc%d" % (octave, size)])
            # Correlation of such rows.  The int8 MFMA contraction holds counts up to 127 -- every count of the 64-region layout
            # MaD.run uses (<= 64 samples per region) -- and refuses larger ones loudly rather than wrapping them.
            if got.max() > 127:
                from mad_amd._lib import MadBackendError
                with pytest.raises(MadBackendError, match="EDOM"):
                    lib.correlate(got, got, 0.95)
                continue
            ph, pl, ps = lib.correlate(got, got, 0.95)      # exact integer dot products over the zero-padded length
            d = got.astype(np.float64)
            nrm = np.sqrt((d * d).sum(1))
            nrm[nrm == 0] = 1.0
            ref = (d @ d.T) / np.outer(nrm, nrm)
            wh, wl = np.nonzero(ref > 0.95)
            np.testing.assert_array_equal(ph, wh)
            np.testing.assert_array_equal(pl, wl)
            np.testing.assert_allclose(ps, ref[wh, wl], rtol=1e-12, atol=0)
        lib.free_field(slot)


def test_the_other_eqsp_sizes(default_lib):
    """Orientator(eqsp_size=16) ("coarse eqsp", BASELINE configs[0]) and Descriptor(subeqsp_size=112) (the reference ships both
    tables, eqsp.py:16) through the drop-in classes against the reference's own outputs (g18): rows, bins, 16-zone histograms,
    the border reject, and the 7 168-count descriptor rows identical; afterwards the default tables are back in place."""
    from mad_amd import synth
    from mad_amd.Descriptor import Descriptor
    from mad_amd.Orientator import Orientator
    g = load("g18_eqsp_sizes.npz")
    lib = default_lib
    for octave in (1, 0):
        field = synth.gradient_field(g["o%d_vol" % octave])
        ms = types.SimpleNamespace(grad_list=[field, field], oct_mode="both", name="g18")
        slot = lib.new_slot()
        lib.upload_field(slot, field)
        ms.device_slots = lambda lib_, s=slot: [s, s]
        coords = g["o%d_coords" % octave]
        k = "o%d_ori16_" % octave
        ori = Orientator(eqsp_size=16)
        rows = ori.assign_orientations(ms, _anchors(coords, octave))
        assert ori.step1_reject == int(g[k + "reject"]) == 1
        np.testing.assert_array_equal([r.index for r in rows], g[k + "anchor"])
        np.testing.assert_array_equal([r.main_bin for r in rows], g[k + "main"])
        np.testing.assert_array_equal([r.sec_bin for r in rows], g[k + "sec"])
        np.testing.assert_array_equal([r.ar_count for r in rows], g[k + "count"])
        np.testing.assert_allclose([r.Rfinal for r in rows], g[k + "R"], rtol=0, atol=1e-14)
        want = g["o%d_dsc112" % octave]
        drows = []
        for c, R in zip(coords[:len(want)], g["o%d_dsc_R" % octave]):
            df = DensityFeature()
            df.set_detector_info(0, octave, [int(v) for v in c], np.zeros(3), np.zeros(3), 1.0)
            df.Rfinal = R
            drows.append(df)
        Descriptor(subeqsp_size=112).generate_descriptors(ms, drows)
        np.testing.assert_array_equal(np.array([r.lin_ar_subeqsp for r in drows]), want)
        # the defaults again: the tables are re-bound by the next default-constructed stage objects
        plain = Orientator().assign_orientations(ms, _anchors(coords, octave))
        assert plain and len(plain[0].ar_count) == 112
        d16 = [DensityFeature() for _ in range(2)]
        for df, c in zip(d16, coords[:2]):
            df.set_detector_info(0, octave, [int(v) for v in c], np.zeros(3), np.zeros(3), 1.0)
            df.Rfinal = np.identity(3)
        Descriptor().generate_descriptors(ms, d16)
        assert len(d16[0].lin_ar_subeqsp) == 1024
        lib.free_field(slot)
