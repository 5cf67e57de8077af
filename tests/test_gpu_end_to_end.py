"""The drop-in flow of run_MaD.py (`from mad import MaD`; add_map / add_subunit / run) on a synthetic dimer:
both planted copies must be found, and the on-disk artefacts must have the reference's layout."""
import csv
import os

import numpy as np
import pytest

from mad_amd import synth

pytestmark = pytest.mark.gpu


def test_run_mad_docks_a_synthetic_dimer(tmp_path, monkeypatch, lib):
    from mad_amd import _lib
    monkeypatch.setattr(_lib, "_default", lib)
    lib._eq_loaded = {}
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(11)
    coords, names, elems = synth.random_globule(1500, 16.0, seed=3)
    synth.write_pdb("subunit.pdb", coords, names, elems)
    truth = [synth.place(coords, synth.random_rotation(rng), t) for t in ([0, 0, 0], [39, 7, -5])]
    synth.write_pdb("assembly.pdb", np.concatenate(truth), names * 2, elems * 2)

    from mad import MaD      # the reference's import line (run_MaD.py:1)
    mad = MaD.MaD()
    mad.add_map("assembly.pdb", 10.0)      # a PDB as map: simulated at 1.2 A/voxel (MaD.py:329-335)
    mad.add_subunit("subunit.pdb", n_copies=2)
    mad.run()

    out = mad.out_folder
    assert out.startswith("results/assembly_subunitx2_res10.000_iso0.000")
    assert os.path.isdir(os.path.join(out, "initial_files")) and os.path.isdir(os.path.join(out, "individual_solutions", "anchor_files"))
    assert os.path.exists(os.path.join(out, "initial_files", "assembly_simulated_map.mrc"))
    assert any(f.endswith((".npz", ".h5")) for f in os.listdir("dsc_db"))
    table = os.path.join(out, "Solutions_refined_subunit.csv")
    with open(table) as fh:
        rows = list(csv.DictReader(fh))
    assert list(rows[0].keys()) == ["ID", "Repeatability", "Weight", "mCC", "RWmCC"]
    assert len(rows) >= 2
    # the two best solutions are the two planted copies (CA RMSD < 3 A), each with a high cross-correlation
    from mad_amd.PDB import PDB
    found = []
    for r in rows[:2]:
        sol = PDB(os.path.join(out, "individual_solutions", "sol_subunit_%s.pdb" % r["ID"]))
        d = [np.sqrt(((sol.coords - t) ** 2).sum(1).mean()) for t in truth]
        found.append(int(np.argmin(d)))
        assert min(d) < 3.0, d
        assert float(r["mCC"]) > 0.8
    assert sorted(found) == [0, 1]
    assert mad.buildable_subunits["subunit"][0] == 2 and len(mad.buildable_subunits["subunit"][1]) >= 2
    # build_assembly (run_MaD.py:76): the best model is the pair of planted copies, clash-free, high CCC
    mad.build_assembly()
    with open(os.path.join(out, "complex_ranking.csv")) as fh:
        models = list(csv.DictReader(fh))
    assert list(models[0].keys()) == ["#", "CC", "Sum(O)", "Std(O)", "Max(O)", "Composition"]
    assert os.path.exists(os.path.join(out, "assembly_models", "Model_1.pdb"))
    assert float(models[0]["Max(O)"]) <= 0.1
    best = max(models, key=lambda r: float(r["CC"]))
    assert float(best["CC"]) > 0.85
    model = PDB(os.path.join(out, "assembly_models", "Model_%s.pdb" % best["#"]))
    assert model.n_atoms == 2 * len(coords) and {r[3] for r in model.info} == {"A", "B"}
    halves = [model.coords[:len(coords)], model.coords[len(coords):]]
    hit = sorted(int(np.argmin([np.sqrt(((h - t) ** 2).sum(1).mean()) for t in truth])) for h in halves)
    assert hit == [0, 1]
    # a second run hits the descriptor cache and lands in a suffixed folder (MaD.py:304-309)
    mad2 = MaD.MaD()
    mad2.add_map("assembly.pdb", 10.0)
    mad2.add_subunit("subunit.pdb", n_copies=2)
    mad2.run()
    assert mad2.out_folder == out + "_1"
    with open(os.path.join(mad2.out_folder, "Solutions_refined_subunit.csv")) as fh:
        rows2 = list(csv.DictReader(fh))
    assert [r["Repeatability"] for r in rows2] == [r["Repeatability"] for r in rows]


def _two_rank_worker(rank, world, port, folder):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MAD_DEVICE="0")
    os.chdir(folder)
    from mad import MaD
    mad = MaD.MaD()
    mad.add_map("assembly.pdb", 10.0)
    mad.add_subunit("subA.pdb")
    mad.add_subunit("subB.pdb")
    mad.run()
    mad.build_assembly()
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_run_mad_on_two_ranks_writes_what_one_rank_writes(tmp_path, monkeypatch, lib):
    """run_MaD.py under a launcher: two ranks (here sharing the one GPU, a gloo control group) deal the structures to describe and
    the subunits to dock; the results folder -- solution tables, solution files, model ranking -- is text for text the one a single
    rank writes (MaD.py:116-190: independent iterations)."""
    import socket
    import torch.multiprocessing as mp
    from mad_amd import _lib
    rng = np.random.default_rng(21)
    parts, all_names, all_elems = [], [], []
    dirs = [str(tmp_path / "two"), str(tmp_path / "one")]
    for d in dirs:
        os.makedirs(d)
    for name, seed, t in (("subA", 3, [0, 0, 0]), ("subB", 5, [40, 5, -6])):
        coords, names, elems = synth.random_globule(1500, 16.0, seed=seed)
        for d in dirs:
            synth.write_pdb(os.path.join(d, name + ".pdb"), coords, names, elems)
        parts.append(synth.place(coords, synth.random_rotation(rng), t))
        all_names += names
        all_elems += elems
    for d in dirs:
        synth.write_pdb(os.path.join(d, "assembly.pdb"), np.concatenate(parts), all_names, all_elems)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, dirs[0]), nprocs=2, join=True)
    monkeypatch.setattr(_lib, "_default", lib)
    lib._eq_loaded = {}
    monkeypatch.chdir(dirs[1])
    from mad import MaD
    mad = MaD.MaD()
    mad.add_map("assembly.pdb", 10.0)
    mad.add_subunit("subA.pdb")
    mad.add_subunit("subB.pdb")
    mad.run()
    mad.build_assembly()
    out = mad.out_folder
    names = ["Solutions_refined_subA.csv", "Solutions_refined_subB.csv", "complex_ranking.csv"]
    names += [os.path.join("individual_solutions", f) for f in sorted(os.listdir(os.path.join(out, "individual_solutions"))) if f.endswith(".pdb")]
    names += [os.path.join("assembly_models", f) for f in sorted(os.listdir(os.path.join(out, "assembly_models"))) if f.endswith(".pdb")]
    assert len(names) >= 6
    for n in names:
        with open(os.path.join(dirs[1], out, n)) as fa, open(os.path.join(dirs[0], out, n)) as fb:
            assert fa.read() == fb.read(), n
