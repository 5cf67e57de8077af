"""Assembly building against the reference's own outputs in tests/golden/g9_assembly.npz:
the overlap restatement of the oracle and the host combinatorics on CPU; the device overlap table, get_overlap and
MaD.build_assembly end to end on the GPU."""
import csv
import io
import os
import types

import numpy as np
import pytest

from mad_amd import assembly, synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g9():
    with np.load(os.path.join(G, "g9_assembly.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _ranking(text):
    rows = list(csv.reader(io.StringIO(str(text))))
    assert rows[0] == assembly.HEADER_CSV
    return [(int(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), r[5]) for r in rows[1:]]


def _write_solutions(g9, folder):
    files = {"A": [], "B": []}
    for key, atoms, names, elems in (("A", g9["atoms_a"], g9["names_a"], g9["elements_a"]), ("B", g9["atoms_b"], g9["names_b"], g9["elements_b"])):
        for i, c in enumerate(atoms):
            files[key].append(os.path.join(folder, "sol_%s_%d.pdb" % (key, i)))
            synth.write_pdb(files[key][-1], c, [str(n) for n in names], [str(e) for e in elems])
    return files


# ------------------------------------------------------------------------------------------- CPU
def test_oracle_overlap_matches_reference(g9):
    from oracle import oracle
    o = g9["lowres_origin"]
    for (i, j, iso, want) in ((0, 3, 1e-8, g9["overlap_all"][0, 3]), (3, 0, 1e-8, g9["overlap_30"]), (0, 3, 0.5, g9["overlap_03_iso"])):
        a, b = g9["lowres_grid_%d" % i].copy(), g9["lowres_grid_%d" % j].copy()
        assert oracle.overlap_ratio(a, o[i], b, o[j], 2, iso) == float(want)
        assert not np.any((a > 0) & (a < iso)) and not np.any((b > 0) & (b < iso))      # clamped in place


def test_oracle_overlap_table_from_atoms(g9):
    """The whole table, with the low-resolution densities simulated by the oracle (a14-a15 restatement)."""
    from oracle import oracle
    sols = [(c, synth.masses([str(e) for e in g9["elements_a"]])) for c in g9["atoms_a"]]
    sols += [(c, synth.masses([str(e) for e in g9["elements_b"]])) for c in g9["atoms_b"]]
    maps = []
    for s, (c, m) in enumerate(sols):
        grid, x0, y0, z0 = oracle.structure_to_density(c, m, 5, 2, isovalue=0.2)
        assert grid.shape == tuple(g9["lowres_dims"][s])
        np.testing.assert_allclose([x0, y0, z0], g9["lowres_origin"][s], atol=1e-9)
        maps.append((grid, (x0, y0, z0)))
    np.testing.assert_array_equal(maps[0][0], g9["lowres_grid_0"])
    n = len(maps)
    table = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            table[i, j] = oracle.overlap_ratio(maps[i][0], maps[i][1], maps[j][0], maps[j][1], 2)
    np.testing.assert_array_equal(table, g9["overlap_all"])


def _half_voxel_cases(g9):
    for dv, want in zip(g9["hv_offsets_voxels"], g9["hv_overlap"]):
        if not np.isnan(want):      # NaN: the reference raises there (boxes of different shapes after rounding)
            yield g9["hv_a"].copy(), g9["hv_origin"], g9["hv_b"].copy(), g9["hv_origin"] + dv * 2.0, float(want)


def test_oracle_overlap_with_origins_half_a_voxel_apart(g9):
    from oracle import oracle
    n = 0
    for a, oa, b, ob, want in _half_voxel_cases(g9):
        assert oracle.overlap_ratio(a, oa, b, ob, 2.0) == want
        n += 1
    assert n >= 8


def test_rank_copies_reproduces_homomer_ranking(g9):
    table = g9["overlap_all"][:5, :5]
    ranked = assembly.rank_copies(table, 2)
    assert len(ranked) == 10
    kept = assembly.select_models(ranked, 10, 0.1)
    want = _ranking(g9["homo_ranking_csv"])
    assert len(kept) == len(want) == 9
    for (idx, s_sum, s_std, s_max), (_, _, w_sum, w_std, w_max, comp) in zip(kept, want):
        assert str([str(i) for i in idx]) == comp
        assert (float(s_sum), float(s_std), float(s_max)) == (w_sum, w_std, w_max)
    assert assembly.rank_copies(table, 1) == [[(s,), 0, 0, 0] for s in range(5)]


def test_subcomplex_names(g9):
    ranked = assembly.rank_copies(g9["overlap_all"][:5, :5], 2)
    names = ["SubComplexA_%i_%s.pdb" % (k, "_".join("A%i" % x for x in idx)) for k, (idx, _, _, mx) in enumerate(ranked) if not mx > 0.1]
    assert names == [str(n) for n in g9["hetero_subcomplexes"] if str(n).startswith("SubComplexA")]


def _table_from_stdout(text, n):
    rows = [l for l in str(text).splitlines() if " | " in l and l.strip()[0].isdigit() and "." in l.split("|")[0]]
    rows = rows[-n:]
    return rows


def test_overlap_table_text(g9):
    got = assembly.format_overlap_table(g9["overlap_all"][:5, :5], ["%i.A" % i for i in range(5)])
    want = [l for l in str(g9["homo_stdout"]).splitlines() if l[:3] in ("0.A", "1.A", "2.A", "3.A", "4.A")]
    assert got == want
    # wide form: parse the reference's 12 x 12 table back and print it again
    want = _table_from_stdout(g9["hetero_stdout"], 12)
    labels = [l.split("|")[0].strip() for l in want]
    vals = np.array([[float(v) for v in l.split("|")[1].split()] for l in want])
    shown = np.where(vals == 0, 0.0, vals)
    got = assembly.format_overlap_table(shown, labels, wide=True)
    assert got == want


def test_rank_models_on_printed_table(g9):
    """_build_models ranks over the sub-complex table; the reference printed it with 3 decimals, which is enough to
    reproduce the order of the all-zero leaders and their compositions."""
    want = _table_from_stdout(g9["hetero_stdout"], 12)
    vals = np.array([[float(v) for v in l.split("|")[1].split()] for l in want])
    ranked = assembly.rank_models(vals, [list(range(9)), [9, 10, 11]])
    kept = assembly.select_models(ranked, 10, 0.1)
    ref = _ranking(g9["hetero_ranking_csv"])
    assert [str([str(i) for i in k[0]]) for k in kept] == [r[5] for r in ref]
    assert all(float(k[1]) == 0.0 for k in kept)


def test_write_complex(tmp_path, g9):
    files = _write_solutions(g9, str(tmp_path))
    out = str(tmp_path / "model.pdb")
    assembly.write_complex([files["A"][0], files["A"][1], files["B"][0]], out)
    from mad_amd.PDB import PDB
    pdb = PDB(out)
    want = np.concatenate([g9["atoms_a"][0], g9["atoms_a"][1], g9["atoms_b"][0]])
    np.testing.assert_array_equal(pdb.coords, want)
    chains = "".join(r[3] for r in pdb.info)
    na, nb = len(g9["names_a"]), len(g9["names_b"])
    assert chains == "A" * na + "B" * na + "C" * nb
    with open(out) as fh:
        assert sum(1 for l in fh if l.startswith("TER")) == 2
    # the reference's Model_1 of the homomer run is solutions 0 + 1
    np.testing.assert_array_equal(g9["homo_assembly_models_Model_1.pdb_coords"], want[:2 * na])
    assert "".join(str(c) for c in g9["homo_assembly_models_Model_1.pdb_chain"]) == "A" * na + "B" * na
    assert int(g9["homo_assembly_models_Model_1.pdb_nter"]) == 1


def test_write_ranking_matches_pandas_text(tmp_path, g9):
    want = str(g9["homo_ranking_csv"])
    rows = [[r[0], np.float32(r[1]), r[2], r[3], r[4], eval(r[5], {})] for r in _ranking(want)]
    path = str(tmp_path / "r.csv")
    assembly.write_ranking(path, rows)
    got = _ranking(open(path).read())
    for a, b in zip(got, _ranking(want)):
        assert a[0] == b[0] and a[2:] == b[2:] and abs(a[1] - b[1]) < 1e-7


def test_score_ensembles_prints_the_reference_ranking(tmp_path, g9, capsys):
    """MaD.score_ensembles (MaD.py:225-287) on the solution tables of a 5-frame ensemble: the printed ranking is the
    reference's, line for line (the bar plot it also saves is visualisation and not reproduced)."""
    from mad_amd.MaD import MaD
    m = MaD()
    m.out_folder = str(tmp_path)
    frames = [str(f) for f in g9["ens_frames"]]
    m.processed_ensembles = {"ens": {f: ["unused.pdb", 1] for f in frames}}
    for f in frames:
        with open(os.path.join(m.out_folder, "Solutions_refined_%s.csv" % f), "w") as fh:
            fh.write("ID,Repeatability,Weight,mCC,RWmCC\n")
            for r in g9["ens_table_" + f]:
                fh.write("%d,%r,%d,%r,%r\n" % (r[0], float(r[1]), int(r[2]), float(r[3]), float(r[4])))
    m.score_ensembles()
    assert capsys.readouterr().out == str(g9["ens_stdout"])


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture()
def default_lib(lib):
    from mad_amd import _lib
    old = _lib._default
    _lib._default = lib
    yield lib
    _lib._default = old


@pytest.mark.gpu
def test_gpu_overlap_matrix_matches_reference(lib, g9):
    coords = list(g9["atoms_a"]) + list(g9["atoms_b"])
    ma, mb = synth.masses([str(e) for e in g9["elements_a"]]), synth.masses([str(e) for e in g9["elements_b"]])
    mass = [ma] * len(g9["atoms_a"]) + [mb] * len(g9["atoms_b"])
    got = lib.overlap_matrix(coords, mass, resolution=5, voxsp=2, density_isovalue=0.2)
    np.testing.assert_array_equal(got, g9["overlap_all"])
    assert lib.overlap_matrix(coords[:1], mass[:1]).shape == (1, 1)
    # a far-away copy: disjoint boxes
    far = lib.overlap_matrix([coords[0], coords[0] + 500.0], [ma, ma])
    assert far[0, 1] == 0.0


@pytest.mark.gpu
def test_gpu_get_overlap(default_lib, g9):
    from mad_amd.structure_utils import get_overlap
    o = g9["lowres_origin"]
    a, b = g9["lowres_grid_0"].copy(), g9["lowres_grid_3"].copy()
    assert get_overlap([a, *o[0]], [b, *o[3]], 2) == float(g9["overlap_all"][0, 3])
    assert get_overlap([b, *o[3]], [a, *o[0]], 2) == float(g9["overlap_30"])
    a64 = g9["lowres_grid_0"].astype(np.float64)
    b2 = g9["lowres_grid_3"].copy()
    assert get_overlap([a64, *o[0]], [b2, *o[3]], 2, isovalue=0.5) == float(g9["overlap_03_iso"])
    assert not np.any((a64 > 0) & (a64 < 0.5)) and not np.any((b2 > 0) & (b2 < 0.5))      # clamped in place, like the reference
    assert get_overlap([a, *o[0]], [b, o[3][0] + 1000.0, o[3][1], o[3][2]], 2) == 0
    for ga, oa, gb, ob, want in _half_voxel_cases(g9):      # python round(): half to even
        assert get_overlap([ga, *oa], [gb, *ob], 2.0) == want
    z = np.zeros((4, 4, 4), np.float32)
    assert get_overlap([z, 0.0, 0.0, 0.0], [b, *o[3]], 2) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("homo", {}), ("hetero", {}), ("hetero_loose", dict(max_models=3, max_overlap_complex=0.5))])
def test_gpu_build_assembly_matches_reference(default_lib, g9, tmp_path, tag, kw, capsys):
    from mad_amd.MaD import MaD
    from mad_amd.PDB import PDB
    files = _write_solutions(g9, str(tmp_path))
    map_sit = str(tmp_path / "g9_map.sit")
    synth.write_situs(map_sit, g9["map_grid"], g9["map_origin"], float(g9["vs"]))
    m = MaD()
    m.out_folder = str(tmp_path / "out")
    os.makedirs(m.out_folder)
    m.processed_map, m.map_name, m.resolution = map_sit, "g9_map", float(g9["res"])
    m.buildable_subunits = {"A": [2, files["A"]]} if tag == "homo" else {"A": [2, files["A"]], "B": [1, files["B"]]}
    m.build_assembly(**kw)
    out = capsys.readouterr().out
    for sub, key in (("subcomplexes", "_subcomplexes"), ("assembly_models", "_models")):
        d = os.path.join(m.out_folder, sub)
        names = sorted(os.listdir(d)) if os.path.isdir(d) else []
        assert names == [str(n) for n in g9[tag + key]]
        for nm in names:
            pdb = PDB(os.path.join(d, nm))
            np.testing.assert_array_equal(pdb.coords, g9["%s_%s_%s_coords" % (tag, sub, nm)])
            assert [r[3] for r in pdb.info] == [str(c) for c in g9["%s_%s_%s_chain" % (tag, sub, nm)]]
    got = _ranking(open(os.path.join(m.out_folder, "complex_ranking.csv")).read())
    want = _ranking(g9[tag + "_ranking_csv"])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a[0] == b[0] and a[2:] == b[2:]
        assert abs(a[1] - b[1]) < 2e-6, (a, b)      # CCC: float32 dots in the reference, float64 accumulation here
    # the overlap tables are printed exactly as the reference prints them
    ref_rows = [l for l in str(g9[tag + "_stdout"]).splitlines() if " | " in l and "." in l.split("|")[0]]
    my_rows = [l for l in out.splitlines() if " | " in l and "." in l.split("|")[0]]
    assert my_rows == ref_rows
