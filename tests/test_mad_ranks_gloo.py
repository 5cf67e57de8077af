"""MaD.run() under a launcher (WORLD_SIZE > 1): the reference's loops over the structures to describe and over the subunits /
frames to dock (MaD.py:116-190) dealt round-robin over the ranks.  Host logic on two gloo ranks, the device stages replaced by
deterministic stand-ins: the results folder, the CSV files and `buildable_subunits` must be those of a one-rank run."""
import os
import socket

import numpy as np

from mad_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs(folder):
    os.makedirs(os.path.join(folder, "ens"), exist_ok=True)
    vol = synth.blob_volume((12, 11, 10), 4, seed=2)
    synth.write_situs(os.path.join(folder, "map.sit"), vol, (1.0, 2.0, 3.0), 2.0)
    for name, seed in (("subA", 1), ("subB", 2), ("subC", 3)):
        c, n, e = synth.random_globule(40, 6.0, seed=seed)
        synth.write_pdb(os.path.join(folder, name + ".pdb"), c, n, e)
    for name, seed in (("f1", 4), ("f2", 5)):
        c, n, e = synth.random_globule(40, 6.0, seed=seed)
        synth.write_pdb(os.path.join(folder, "ens", name + ".pdb"), c, n, e)


def _fake_mad():
    from mad_amd.DensityFeature import DensityFeature
    from mad_amd.MaD import MaD

    class FakeMaD(MaD):
        def _describe_struct(self, struct, *a):      # three rows whose content depends on the structure's file
            seed = sum(os.path.basename(struct).encode())
            rng = np.random.default_rng(seed)
            rows = []
            for i in range(3):
                df = DensityFeature()
                df.set_from_file_dsc(i, 1 + i, 2 + i, 1, 112, 16, np.array([1.0, 2, 3]) * i, rng.normal(size=3), rng.normal(size=3), np.eye(3),
                                     rng.integers(0, 60, 1024).astype(np.int16))
                rows.append(df)
            return rows

        def _match_filter_refine(self, pdbfile, n_copies, k, cc_threshold, weight_threshold, n_samples):
            hi = self._load_descriptors(self.dsc_dict[k]) if isinstance(self.dsc_dict[k], str) else self.dsc_dict[k]
            score = float(np.sum([df.lin_ar_subeqsp.sum() for df in hi]) + np.sum([df.lin_ar_subeqsp.sum() for df in self.map_dsc]))
            os.makedirs(os.path.join(self.out_folder, "individual_solutions"), exist_ok=True)
            files = []
            for i in range(2):
                f = os.path.join(self.out_folder, "individual_solutions", "sol_%s_%i.pdb" % (k, i))
                with open(f, "w") as fh:
                    fh.write("REMARK %s %d %r\n" % (k, n_copies, score))
                files.append(f)
            with open(os.path.join(self.out_folder, "Solutions_refined_%s.csv" % k), "w") as fh:
                fh.write("ID,Repeatability,Weight,mCC,RWmCC\n0,%r,%d,0.5,%r\n" % (score, n_copies, score / 2))
            return files if k != "subB" else []      # a subunit without solutions is left out of buildable_subunits (MaD.py:169)

    return FakeMaD


def _run(folder, inputs):
    cwd = os.getcwd()
    os.chdir(folder)
    try:
        mad = _fake_mad()()
        mad.add_map(os.path.join(inputs, "map.sit"), 8.0)
        mad.add_subunit(os.path.join(inputs, "subA.pdb"), n_copies=2)
        mad.add_subunit(os.path.join(inputs, "subB.pdb"))
        mad.add_subunit(os.path.join(inputs, "subC.pdb"))
        mad.add_subunit(os.path.join(inputs, "ens"), n_copies=1)
        mad.run()
        return mad
    finally:
        os.chdir(cwd)


def _worker(rank, world, port, folder, inputs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    mad = _run(folder, inputs)
    np.save(os.path.join(folder, "buildable%d.npy" % rank), np.array([repr(mad.buildable_subunits), mad.out_folder], dtype=object), allow_pickle=True)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            p = os.path.join(d, f)
            if f.endswith((".csv", ".pdb")) and "initial_files" not in p:
                with open(p) as fh:
                    out[os.path.relpath(p, root)] = fh.read()
    return out


def test_run_on_two_ranks_equals_one_rank(tmp_path):
    import torch.multiprocessing as mp
    inputs = str(tmp_path / "inputs")
    _make_inputs(inputs)
    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    os.makedirs(two)
    os.makedirs(one)
    mp.spawn(_worker, args=(2, _free_port(), two, inputs), nprocs=2, join=True)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    ref = _run(one, inputs)
    got = [np.load(os.path.join(two, "buildable%d.npy" % r), allow_pickle=True) for r in range(2)]
    assert got[0][1] == got[1][1] == ref.out_folder                      # one results folder, the reference's name
    assert got[0][0] == got[1][0] == repr(ref.buildable_subunits)       # same keys, same order, same files on every rank
    assert list(ref.buildable_subunits) == ["subA", "subC", "ens"]
    t2, t1 = _tree(os.path.join(two, ref.out_folder)), _tree(os.path.join(one, ref.out_folder))
    assert sorted(t2) == sorted(t1) and len(t1) >= 5 + 4 * 2
    for name in t1:
        assert t2[name] == t1[name], name
    # every structure was described exactly once, by somebody: the cache holds the map, three subunits and two frames
    assert len(os.listdir(os.path.join(two, "dsc_db"))) == len(os.listdir(os.path.join(one, "dsc_db"))) == 6


def _failing_worker(rank, world, port, folder, inputs):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MAD_DIST_TIMEOUT_S="120")
    Fake = _fake_mad()

    class Failing(Fake):
        def _match_filter_refine(self, pdbfile, n_copies, k, *a):
            if k == "subA":      # rank 0's first docking dies the way the reference's helpers do (print + sys.exit(1), PDB.py:13-15)
                raise SystemExit(1)
            return super()._match_filter_refine(pdbfile, n_copies, k, *a)

    cwd = os.getcwd()
    os.chdir(folder)
    t0 = time.time()
    try:
        mad = Failing()
        mad.add_map(os.path.join(inputs, "map.sit"), 8.0)
        mad.add_subunit(os.path.join(inputs, "subA.pdb"), n_copies=2)
        mad.add_subunit(os.path.join(inputs, "subB.pdb"))
        try:
            mad.run()
            outcome = "finished"
        except RuntimeError as e:
            outcome = "raised: %s" % e
    finally:
        os.chdir(cwd)
    with open(os.path.join(folder, "outcome%d.txt" % rank), "w") as fh:
        fh.write("%s\n%.1f\n" % (outcome, time.time() - t0))


def test_a_rank_that_fails_stops_every_rank(tmp_path):
    """A rank that raises (or sys.exits, as the reference's helpers do on bad input) inside its share of MaD.run() must not leave
    the others in a barrier until the group's timeout: the ranks agree on the outcome before every exchange, and all of them raise."""
    import torch.multiprocessing as mp
    inputs = str(tmp_path / "inputs")
    _make_inputs(inputs)
    two = str(tmp_path / "two")
    os.makedirs(two)
    mp.spawn(_failing_worker, args=(2, _free_port(), two, inputs), nprocs=2, join=True)
    for r in range(2):
        with open(os.path.join(two, "outcome%d.txt" % r)) as fh:
            outcome, secs = fh.read().splitlines()
        assert outcome.startswith("raised: MaD> rank 0 failed (SystemExit"), (r, outcome)
        assert float(secs) < 60.0
