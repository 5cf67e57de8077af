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


def test_run_mad_on_the_frozen_c1_workload(tmp_path, monkeypatch, lib):
    """BASELINE configs[0] as frozen in bench/configs/c1.json -- the 64^3 map at 2.0 A per voxel of two distinct subunits (seeds 1,
    2), 10 A -- through the run_MaD.py flow (run_MaD.py:64-76): the map from a file, both subunits docked, each one's best solution
    on its planted copy, build_assembly's best model = the planted dimer.  (run(ori_eqsp_size=16), the survey's "coarse eqsp", is
    accepted and ignored exactly as the reference ignores it: MaD.py:87)"""
    import bench
    from mad_amd import _lib, mapio
    from mad_amd.PDB import PDB
    monkeypatch.setattr(_lib, "_default", lib)
    lib._eq_loaded = {}
    monkeypatch.chdir(tmp_path)
    W = bench.WORKLOADS["c1"]
    the_map, subs, _ = bench.build_inputs(lib, W, 0)
    assert the_map.shape == (64, 64, 64) and len(subs) == 2
    mapio.write_mrc("c1map.mrc", the_map.grid, the_map.origin, W["vs"])
    for s, seed in enumerate(W["seeds"]):
        coords, names, elems = synth.random_globule(W["n_atoms"], W["radius"], seed=seed)
        np.testing.assert_array_equal(coords, subs[s].atoms)
        synth.write_pdb("sub%d.pdb" % s, coords, names, elems)
    truth = the_map.placed
    for st_ in [the_map] + subs:
        st_.ms.release_device()

    from mad import MaD
    mad = MaD.MaD()
    mad.add_map("c1map.mrc", W["res"])
    mad.add_subunit("sub0.pdb")
    mad.add_subunit("sub1.pdb")
    mad.run(ori_eqsp_size=16)
    out = mad.out_folder
    for s in range(2):
        with open(os.path.join(out, "Solutions_refined_sub%d.csv" % s)) as fh:
            rows = list(csv.DictReader(fh))
        assert list(rows[0].keys()) == ["ID", "Repeatability", "Weight", "mCC", "RWmCC"] and len(rows) >= 1
        sol = PDB(os.path.join(out, "individual_solutions", "sol_sub%d_%s.pdb" % (s, rows[0]["ID"])))
        rmsd = float(np.sqrt(((sol.coords - truth[s]) ** 2).sum(1).mean()))
        print("c1 subunit %d: %d solutions, best: repeatability %s, mCC %s, RMSD to the planted copy %.2f A" % (s, len(rows), rows[0]["Repeatability"], rows[0]["mCC"], rmsd))
        assert rmsd < 4.0 and float(rows[0]["mCC"]) > 0.7      # a voxel is 2 A, the map 10 A
    mad.build_assembly()
    with open(os.path.join(out, "complex_ranking.csv")) as fh:
        models = list(csv.DictReader(fh))
    best = max(models, key=lambda r: float(r["CC"]))
    print("c1 assembly: %d models, best CC %s" % (len(models), best["CC"]))
    assert float(best["CC"]) > 0.8


@pytest.mark.parametrize("overflow", [False, True], ids=["sized", "image_overflows"])
def test_bench_on_two_ranks_shares_the_map_build_and_gathers_the_topk(tmp_path, overflow):
    """`python bench.py --gpus 2 --workload c4` as the driver starts it (the parent spawns its ranks with torch.distributed.run), here
    with the collectives over gloo because the box has one GPU (MAD_DIST_BACKEND=gloo: both ranks share it).  The N > 1 path end to
    end in fresh child processes: the map set assembled from the two ranks' shares is bit for bit the set one GPU builds
    (ShardedSetBuild: export -> all-gather -> import), and every subunit's top-k after the exchange equals the one-GPU run's.
    image_overflows: one group's wire images are cut to 64 rows after the set-up (--rehearse-resize), so an import inside the pipelined
    steps reports MAD_ENOSPC; the ranks learn of it through the top-k exchange, drop what is in flight, size the images again
    together and start over -- same checks, one re-sizing on the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAD_DIST_BACKEND="gloo", MAD_DIST_TIMEOUT_S="240", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c4", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"] +
                       (["--rehearse-resize", "64"] if overflow else []), cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["subunits"] == 8 and d["config"]["subunits_this_rank"] == 4
    assert d["config"]["sharded_map_set_identical_to_unsharded"] is True
    assert d["one_gpu_same_workload"]["topk_identical_to_sharded_run"] is True
    assert len(d["config"]["correlations_per_step_by_rank"]) == 2 and min(d["config"]["correlations_per_step_by_rank"]) > 0
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["map_image_resizes"] == (1 if overflow else 0)
