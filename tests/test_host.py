"""Host-side logic without a GPU: the C-ABI library loads and exports every symbol the header
declares, the file formats round-trip, and the host stages either side of the hot path
(MapSpace, Detector, _filter_dsc_pairs, Kabsch) reproduce the reference's golden outputs."""
import os
import re

import numpy as np
import pytest

from mad_amd import _lib, mapio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def load(name):
    with np.load(os.path.join(G, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mad_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mad_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 30
    assert sorted(_lib.SYMBOLS) == declared, "mad_amd/_lib.py SYMBOLS and include/mad_amd.h disagree"
    dll = _lib.load_library()
    missing = [s for s in declared if not hasattr(dll, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.MadBackendError, match="MI355X"):
        _lib.Lib(0)


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "mad_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|mad_oracle|libmad_oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, "product files reference the test oracle: %s" % bad


def test_volume_formats_round_trip(tmp_path):
    vol = synth.blob_volume((11, 13, 9), 5, 3)
    org = (-12.5, 3.25, 40.0)
    p = str(tmp_path / "v.mrc")
    mapio.write_mrc(p, vol, org, 1.5)
    grid, vs, o, dims = mapio.load_mrc_as_xyz(p)
    np.testing.assert_array_equal(grid, vol)
    assert abs(vs - 1.5) < 1e-6 and dims == vol.shape
    assert o == (-12.0, 3.0, 40.0)      # the reference truncates MRC origins to integers (Dmap.py:38)
    s = str(tmp_path / "v.sit")
    mapio.write_situs(s, vol, org, 1.5)
    g2, vs2, o2 = mapio.read_situs(s)
    assert g2.shape == vol.shape and vs2 == 1.5 and o2 == org
    np.testing.assert_allclose(g2, vol, atol=5e-7)


def test_pdb_round_trip(tmp_path):
    from mad_amd.PDB import PDB
    coords, names, elems = synth.random_globule(50, 8.0, 2)
    p = str(tmp_path / "a.pdb")
    synth.write_pdb(p, coords, names, elems)
    pdb = PDB(p)
    np.testing.assert_array_equal(pdb.coords, coords)
    assert pdb.CA_idx == tuple(range(1, 50, 4)) and pdb.n_atoms == 50
    assert [r[5] for r in pdb.info[:4]] == ["N", "C", "C", "O"]
    q = str(tmp_path / "b.pdb")
    pdb.rotate_atoms(np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]]))
    pdb.translate_atoms([1, 2, 3])
    pdb.write_pdb(q)
    np.testing.assert_allclose(PDB(q).coords, pdb.coords, atol=5e-4)
    np.testing.assert_allclose(pdb.atom_masses()[:4], [14.0067, 12.011, 12.011, 15.9994])


def test_math_utils_against_reference():
    from mad_amd.math_utils import euler_rod_mat, get_rototrans_SVD, unit_vector
    g = load("g1_eqsp_math.npz")
    for a, t, m in zip(g["rod_axes"], g["rod_angles"], g["rod_mats"]):
        np.testing.assert_array_equal(euler_rod_mat(unit_vector(a), t), m)
    R, T = get_rototrans_SVD(g["kabsch_mobile"], g["kabsch_reference"])
    np.testing.assert_allclose(R, g["kabsch_R"], atol=1e-14)
    np.testing.assert_allclose(T, g["kabsch_T"], atol=1e-12)


def test_scale_space_checker_against_reference():
    """oracle/scale_space.py (the scipy restatement that checks the device MapSpace) vs the reference's own output."""
    from oracle import scale_space as OS
    g = load("g_mapspace.npz")
    grid = g["map_grid"].astype(np.float64)
    grid = grid / np.amax(grid).astype(np.float32)      # what MapSpace.py:96 does to a situs map
    vol = OS.build_volumes(grid, pad=9, oct_mode="both", sig_init=2.0, sig_presmooth=1)
    for o in (0, 1):
        idx = g["grad_idx_%d" % o]
        assert tuple(g["grad_shape_%d" % o]) == vol["grad_list"][o].shape
        # the fixture's map went through a 6-decimal situs text file; this one did not
        np.testing.assert_allclose(vol["grad_list"][o][idx[:, 0], idx[:, 1], idx[:, 2]], g["grad_val_%d" % o], rtol=0, atol=2e-6)
        np.testing.assert_allclose(vol["map_space"][o][idx[:, 0], idx[:, 1], idx[:, 2]], g["log_val_%d" % o], rtol=0, atol=2e-6)


def test_scale_tables_are_scipys():
    """The weights / spline operator handed to mad_space_build are the ones scipy uses."""
    from scipy.interpolate import interp1d
    from scipy.ndimage import _filters
    from mad_amd import scale_tables as st
    for sig in (1, 2, 2.0, 1.5, 3):
        r = st.kernel_radius(sig)
        for order in (0, 2):
            np.testing.assert_array_equal(st.gaussian_kernel1d(sig, order, r), _filters._gaussian_kernel1d(sig, order, r))
    rng = np.random.default_rng(5)
    for n in (4, 5, 7, 27, 83):
        y = rng.random((n, 3))
        ref = interp1d(np.arange(n), y, axis=0, kind="cubic")(np.arange(0, n - 0.5, 0.5))
        np.testing.assert_allclose(st.spline_apply(y, 0), ref, rtol=0, atol=2e-15)
    with pytest.raises(ValueError):
        st.spline_tables(3)


def test_nearest_gradient_lookup():
    """The stand-in for RegularGridInterpolator(method="nearest"): tie rule and bounds of the reference's interpolator."""
    from mad_amd.MapSpace import NearestGradient
    vals = np.random.default_rng(2).random((8, 9, 10, 3))
    rgi = NearestGradient(vals)
    np.testing.assert_array_equal(rgi(np.array([[3.5, 4.5, 5.5], [3.51, 4.0, 5.0]])), vals[[3, 4], [4, 4], [5, 5]])
    with pytest.raises(ValueError):
        rgi(np.array([[-0.1, 1, 1]]))


def test_filter_dsc_pairs_against_reference(tmp_path):
    from mad_amd.MaD import MaD
    g4, g5, g7 = load("g4_match.npz"), load("g5_filter.npz"), load("g7_density_ccc.npz")
    names = [synth.ATOM_CYCLE[i % 4][0] for i in range(len(g5["atoms"]))]
    pdbfile = str(tmp_path / "sub.pdb")
    synth.write_pdb(pdbfile, g5["atoms"], names, [str(e) for e in g7["elements"]])
    filt = MaD()._filter_dsc_pairs(pdbfile, list(g4["results"]), g4["lo_cloud"], g4["hi_cloud"], wthresh=4, n_samples=120)
    assert len(filt) == int(g5["n"]) >= 1
    np.testing.assert_array_equal([f[4] for f in filt], g5["weight"])
    np.testing.assert_array_equal([f[5] for f in filt], g5["repeat"])
    np.testing.assert_array_equal([f[3] for f in filt], g5["cc"])
    np.testing.assert_allclose([f[2] for f in filt], g5["R"], atol=0)
    np.testing.assert_allclose([f[7].coords for f in filt], g5["placed"], rtol=0, atol=1e-10)


def test_descriptor_cache_round_trip(tmp_path, monkeypatch):
    from mad_amd.DensityFeature import DensityFeature
    from mad_amd.MaD import MaD
    g = load("g4_match.npz")
    rows = []
    for i in range(10):
        df = DensityFeature()
        df.set_from_file_dsc(int(g["hi_index"][i]), int(g["hi_main"][i]), int(g["hi_sec"][i]), int(g["hi_oct"][i]), 112, 16,
                             g["hi_coords"][i].astype(float), g["hi_subv"][i], g["hi_subv"][i], g["hi_R"][i], g["hi_dsc"][i])
        rows.append(df)
    m = MaD()
    name = str(tmp_path / "x_res8.0.h5")
    m._save_descriptors(rows, name)
    assert m._cache_exists(name)
    back = m._load_descriptors(name)
    assert len(back) == 10
    for a, b in zip(rows, back):
        assert (a.index, a.main_bin, a.sec_bin, a.oct_scale) == (b.index, b.main_bin, b.sec_bin, b.oct_scale)
        np.testing.assert_array_equal(a.lin_ar_subeqsp, b.lin_ar_subeqsp)
        np.testing.assert_array_equal(a.Rfinal, b.Rfinal)
        np.testing.assert_array_equal(a.subv_map_coords, b.subv_map_coords)


def test_pdb_reader_and_writer_on_awkward_records(tmp_path):
    """mad_amd.PDB against the reference's PDB (PDB.py:19-97) on a file with 4-letter atom names, HETATM, a missing element
    column, unparsable serial / residue numbers (the previous atom's values are kept), TER / ANISOU lines, touching
    coordinate columns and a truncated line: parsed fields, CA / backbone indices and the written file, text for text."""
    import os
    from mad_amd.PDB import PDB
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_pdb.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    src = str(tmp_path / "torture.pdb")
    with open(src, "w") as fh:
        fh.write(str(g["text"]))
    pdb = PDB(src)
    assert pdb.n_atoms == int(g["n_atoms"])
    np.testing.assert_array_equal(pdb.coords, g["coords"])
    assert [r[0] for r in pdb.info] == [int(v) for v in g["serial"]]
    assert [r[4] for r in pdb.info] == [int(v) for v in g["resnum"]]
    for col, key in ((1, "name"), (2, "resname"), (3, "chain"), (5, "element"), (6, "record")):
        assert [r[col] for r in pdb.info] == [str(v) for v in g[key]], key
    assert tuple(pdb.CA_idx) == tuple(int(v) for v in g["ca_idx"]) and list(pdb.BB_idx) == [int(v) for v in g["bb_idx"]]
    out = str(tmp_path / "out.pdb")
    pdb.write_pdb(out)
    assert open(out).read() == str(g["written"])


def test_dmap_on_a_situs_file_against_the_reference(tmp_path, capsys):
    """mad_amd.Dmap against the reference's Dmap (Dmap.py:7-97, 377-390) on the same Situs file: isovalue + normalisation,
    padding, an isovalue above the maximum (warning, then 0), reduce_void, and the Situs text written back."""
    import os
    from mad_amd.Dmap import Dmap
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_dmap.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    src = str(tmp_path / "g13.sit")
    with open(src, "w") as fh:
        fh.write(str(g["sit_text"]))

    def geo(d):
        return np.array([d.xi, d.yi, d.zi, d.voxsp, d.xb, d.yb, d.zb], dtype=np.float64)

    for tag, kw in (("plain", {}), ("iso", dict(isovalue=0.8)), ("raw", dict(isovalue=0.3, normalize=False, pad=3)), ("huge_iso", dict(isovalue=50.0))):
        d = Dmap(src, **kw)
        assert d.grid3d.dtype == g[tag + "_grid"].dtype, tag
        np.testing.assert_array_equal(d.grid3d, g[tag + "_grid"], err_msg=tag)
        np.testing.assert_array_equal(geo(d), g[tag + "_geo"], err_msg=tag)
        if tag == "iso":
            d.reduce_void(zeros_padding=4)
            np.testing.assert_array_equal(d.grid3d, g["reduced_grid"])
            np.testing.assert_array_equal(geo(d), g["reduced_geo"])
            out = str(tmp_path / "out.sit")
            d.write_to_sit(out)
            same_text = open(out).read() == str(g["reduced_sit_text"])      # not inside the assert: pytest would diff 100 KB strings
            assert same_text
    assert "larger than maximum density" in capsys.readouterr().out


def test_save_solutions_refined_against_the_reference(tmp_path, capsys):
    """MaD._save_solutions_refined (MaD.py:923-958) on three made-up solutions: the printed table, Solutions_refined_<key>.csv as
    pandas writes it, the solution PDBs and the corresp_anchors PDBs, text for text."""
    import os
    from mad_amd.MaD import MaD
    from mad_amd.PDB import PDB
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_solutions_io.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    src = str(tmp_path / "sub.pdb")
    with open(src, "w") as fh:
        fh.write(str(g["sub_text"]))
    m = MaD()
    m.out_folder = str(tmp_path / "out")
    os.makedirs(m.out_folder)
    sols = []
    for i in range(3):
        pdb = PDB(src)
        pdb.set_coords(g["coords_%d" % i])
        repeat, weight, ccc, score = g["row_%d" % i]
        sols.append([pdb, g["corresp_%d" % i], float(repeat), int(weight), np.float32(ccc), [], float(score)])
    files = m._save_solutions_refined(sols, "subA")
    assert [os.path.basename(f) for f in files] == [str(f) for f in g["files"]]
    out = capsys.readouterr().out
    same = [out == str(g["stdout"]), open(os.path.join(m.out_folder, "Solutions_refined_subA.csv")).read() == str(g["csv"])]
    for i in range(3):
        same.append(open(os.path.join(m.out_folder, "individual_solutions", "sol_subA_%d.pdb" % i)).read() == str(g["sol_pdb_%d" % i]))
        same.append(open(os.path.join(m.out_folder, "individual_solutions", "anchor_files", "corresp_anchors_subA_%d.pdb" % i)).read()
                    == str(g["corresp_pdb_%d" % i]))
    assert same == [True] * len(same), same


def test_move_structure_against_the_reference(tmp_path):
    """structure_utils.move_structure / move_copy_structure (structure_utils.py:8-57): the PDB files they write."""
    import os
    from mad_amd.structure_utils import move_copy_structure, move_structure
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_move_structure.npz"), allow_pickle=False) as z:
        g = {k: str(z[k]) for k in z.files}
    src = str(tmp_path / "g15.pdb")
    with open(src, "w") as fh:
        fh.write(g["src"])
    same = [open(move_structure(src)).read() == g["moved_default"],
            open(move_structure(src, t=[1.5, -2.0, 3.25], a=0.1, b=-0.2, c=0.3, suffix="_b")).read() == g["moved_t"]]
    for tag, kw in (("copy", {}), ("copy_transform", dict(transform=True)), ("copy_transform_t0", dict(transform=True, t=[]))):
        dst = str(tmp_path / ("%s.pdb" % tag))
        move_copy_structure(src, dst, **kw)
        same.append(open(dst).read() == g[tag])
    assert same == [True] * 5, same


def test_check_localize_against_the_reference():
    """Detector.check_localize (Detector.py:53-123): verdict, voxel and sub-voxel position for every local maximum of two smooth
    volumes and for 150 off-peak starts each, against the reference's results (the peak search in front of it stays unpinned:
    scikit-image is not available)."""
    import os
    from mad_amd.Detector import Detector
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_localize.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    det = Detector()
    for tag in ("f32", "f64"):
        vol = g[tag + "_vol"]
        n_good = 0
        for p_, ok, cc, sc in zip(g[tag + "_cand"], g[tag + "_good"], g[tag + "_coord"], g[tag + "_sub"]):
            got_ok, got_c, got_s = det.check_localize(vol, np.array(p_))
            assert bool(got_ok) == bool(ok), (tag, p_)
            assert [int(v) for v in got_c] == [int(v) for v in cc], (tag, p_)
            np.testing.assert_array_equal(np.array([float(v) for v in got_s]), sc)
            n_good += bool(ok)
        assert n_good >= 40


def test_density_upload_cache_follows_the_grid_object_and_its_contents():
    """structure_utils keeps ONE map on the device for refinement / CCC.  It must be re-uploaded when another grid object
    comes along (even one that recycles an id), when the same object is edited in place, or when its placement changes --
    and only then."""
    from mad_amd import structure_utils as su

    class FakeLib(object):
        ctx = 1

        def __init__(self):
            self.uploads = 0

        def upload_density(self, grid, origin, voxsp):
            self.uploads += 1

    class FakeMap(object):
        def __init__(self, grid):
            self.grid3d, self.xi, self.yi, self.zi, self.voxsp = grid, 1.0, 2.0, 3.0, 1.5

    rng = np.random.default_rng(0)
    lib, m = FakeLib(), FakeMap(rng.random((40, 41, 42)).astype(np.float32))
    su.invalidate_density()
    su._ensure_density(lib, m)
    su._ensure_density(lib, m)
    assert lib.uploads == 1
    m.grid3d[m.grid3d < 0.5] = 0      # thresholded in place
    su._ensure_density(lib, m)
    assert lib.uploads == 2
    m.grid3d *= 0.5                   # normalised in place
    su._ensure_density(lib, m)
    assert lib.uploads == 3
    m2 = FakeMap(m.grid3d.copy())     # another object with the same contents and placement
    su._ensure_density(lib, m2)
    assert lib.uploads == 4
    m2.xi += 1.5
    su._ensure_density(lib, m2)
    assert lib.uploads == 5
    su._ensure_density(FakeLib(), m2)      # another context: nothing of this one is valid there
    su._ensure_density(lib, m2)
    assert lib.uploads == 6
    su.invalidate_density()


def test_table_classifier_of_the_compact_texels_is_conservative():
    """Host model of k_describe's first tier (4-byte texels: a unit direction in 3 x 10 bits, classified through the conservative
    belt / pseudo-angle tables of mad_set_eqsp): over random directions and rotations, whatever the table decides is the zone of
    the exact float64 classification of the unquantised direction; 3-5 % stay undecided and go on to the exact tiers."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("check_tab", os.path.join(ROOT, "tools", "check_tab_classifier.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = ["check_tab_classifier.py", "400000"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = old
