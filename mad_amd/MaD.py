"""Orchestrator with the reference's entry points (mad/MaD.py:25-99):

    mad = MaD.MaD(); mad.add_map(path, resolution); mad.add_subunit(path, n_copies=k); mad.run()

On the hot path: `_describe_struct` (orientation + description on
the GPU), `_match_dsc` (int8-MFMA correlation, pose scoring, top-k on the GPU) and
`_refine_filtered_solutions` (batched persistent refinement, density simulation and CCC
on the GPU).  `_filter_dsc_pairs` (the greedy cloud clustering that picks which poses
get refined, MaD.py:456-553) is the "next" row and runs on the host in float64 numpy.

Several GPUs (round 3): started under a launcher (`torchrun --nproc-per-node N run_MaD.py ...`: WORLD_SIZE / RANK / LOCAL_RANK in
the environment), `run()` deals the structures to describe and then the subunits / ensemble frames to dock round-robin over the
ranks -- the reference's loops over them are independent iterations (MaD.py:116-190) -- every rank works on its GPU, the
solution files land in the one results folder rank 0 created, and the lists of solutions are all-gathered so that every rank
ends with the reference's `buildable_subunits`; `build_assembly` runs on rank 0.  The exchanges are a few python objects over a
gloo group (names, file lists): control plane, nothing of the data path crosses ranks.

Kept from the reference: method names, argument order and defaults, the
`results/<map>_<comps>_res..._iso...` folder layout, the `Solutions_refined_<k>.csv`
header and the descriptor cache file name (MaD.py:118).  Replaced: the h5py cache
becomes an `.npz` with the same four datasets when h5py is missing; the map is written
with the built-in MRC writer.  Assembly building (MaD.py:632-843) is out of scope for
the hot path; `build_assembly` is provided by mad_amd.assembly when present.
"""
import os
from copy import deepcopy
from operator import itemgetter

import numpy as np

from . import _lib
from .DensityFeature import DensityFeature
from .Descriptor import Descriptor
from .Detector import Detector
from .Dmap import Dmap
from .MapSpace import MapSpace
from .math_utils import get_rototrans_SVD
from .Orientator import Orientator
from .PDB import PDB
from .structure_utils import dock_refine_score_many, move_copy_structure

try:      # optional: the reference's cache format
    import h5py
except Exception:      # pragma: no cover - h5py is absent in the build image
    h5py = None

# result-row columns of MaD.py:451 / :461-474
D_CC, REPEAT, LO_IDX, LO_OCT, LO_BIN, HI_IDX, HI_OCT, HI_BIN = range(8)
HI_COORD, LO_COORD = slice(8, 11), slice(11, 14)
R1, R2, R3 = slice(14, 17), slice(17, 20), slice(20, 23)


class _RowSet(object):
    """Host view of a descriptor list + its device-resident twin (mad_set)."""

    def __init__(self, lib, dsc_list):
        n = len(dsc_list)
        self.n_rows = n
        subv = np.array([df.subv_map_coords for df in dsc_list], dtype=np.float64).reshape(-1, 3)
        # anchors = unique sub-voxel coordinates, lexicographically sorted like np.unique (MaD.py:427-428)
        if n:
            self.anchors, self.row_anchor = np.unique(subv, axis=0, return_inverse=True)
            self.row_anchor = np.asarray(self.row_anchor, dtype=np.int32).reshape(-1)
        else:
            self.anchors, self.row_anchor = np.zeros((0, 3)), np.zeros(0, np.int32)
        first = np.zeros(len(self.anchors), np.int64)
        first[self.row_anchor[::-1]] = np.arange(n)[::-1]
        idx = np.array([df.index for df in dsc_list], dtype=np.int32)
        octv = np.array([df.oct_scale for df in dsc_list], dtype=np.int32)
        main = np.array([df.main_bin for df in dsc_list], dtype=np.int32)
        R = np.array([df.Rfinal for df in dsc_list], dtype=np.float64).reshape(-1, 9)
        dsc = np.array([df.lin_ar_subeqsp for df in dsc_list], dtype=np.int16).reshape(n, -1)
        self.dev = lib.set_load(self.row_anchor, main, R, dsc, self.anchors, idx[first] if n else idx, octv[first] if n else octv)


class MaD(object):
    def __init__(self):
        self.input_map = None
        self.input_subunits = {}
        self.input_ensembles = {}
        self.processed_map = None
        self.processed_subunits = {}
        self.processed_ensembles = {}
        self.buildable_subunits = {}
        self.out_folder = None
        self.dsc_dict = {}
        self._rowsets = {}

    # ------------------------------------------------------------------ inputs
    def add_subunit(self, sub_filename_folder, n_copies=1, identifier=""):
        assert os.path.exists(sub_filename_folder), "MaD> subunit or ensemble not found: %s" % sub_filename_folder
        if os.path.isfile(sub_filename_folder):
            filename = os.path.splitext(os.path.split(sub_filename_folder)[-1])[0]
            key = identifier if identifier != "" else filename
            if key in self.input_subunits:
                print("MaD> subunit %s already added; overwriting" % filename)
            self.input_subunits[key] = [sub_filename_folder, n_copies]
            print("MaD> Added: subunit %s" % sub_filename_folder)
        elif os.path.isdir(sub_filename_folder):
            folder = os.path.basename(os.path.normpath(sub_filename_folder))
            key = identifier if identifier != "" else folder
            frames = [os.path.join(sub_filename_folder, x) for x in os.listdir(sub_filename_folder) if x.split(".")[-1] in ("pdb", "PDB")]
            if not frames:
                print("MaD> No PDB files found in ensemble folder %s" % sub_filename_folder)
                return
            self.input_ensembles[key] = {}
            for frame in frames:
                fkey = os.path.splitext(os.path.split(frame)[-1])[0]
                print("      > Added frame: ", fkey)
                self.input_ensembles[key][fkey] = [frame, n_copies]
            print("MaD> Added: ensemble %s of %i frames" % (key, len(frames)))
        else:
            print("MaD> Error: %s not a valid structure or ensemble" % sub_filename_folder)

    def add_map(self, input_map, resolution, isovalue=0):
        assert os.path.exists(input_map), "MaD> Exp. map not found: %s" % input_map
        assert resolution > 0, "MaD> Map cannot have a negative resolution"
        self.resolution = resolution
        self.isovalue = isovalue
        self.input_map = input_map
        self.map_name = os.path.splitext(os.path.split(input_map)[-1])[0]
        print("MaD> Added: density map %s, resolution %.2f A" % (self.map_name, self.resolution))

    # ------------------------------------------------------------------ ranks
    def _ranks(self):
        """(rank, world, torch.distributed or None).  More than one rank only under a launcher (WORLD_SIZE > 1)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1:
            return 0, 1, None
        import torch.distributed as dist
        if not dist.is_initialized():
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29544")      # (a launcher sets its own; this is the fallback of a hand-started job)
            # a rank that died must not hold the others for the backend's default half hour
            dist.init_process_group(os.environ.get("MAD_CONTROL_BACKEND", "gloo"),
                                    timeout=datetime.timedelta(seconds=int(os.environ.get("MAD_DIST_TIMEOUT_S", "600"))))
        return dist.get_rank(), dist.get_world_size(), dist

    def _all_ranks(self, dist, work):
        """Runs `work()` on this rank and agrees with the others on the outcome BEFORE the exchange that follows: the helpers of the
        reference print and sys.exit(1) on bad input (Dmap.py:45-49, PDB.py:13-15), and a rank that leaves alone would hold the
        others in the next barrier until the group's timeout.  Every rank raises when any rank failed."""
        err, out = None, None
        try:
            out = work()
        except BaseException as e:      # SystemExit included
            err = "%s: %s" % (type(e).__name__, e)
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, err)
        bad = [(r, e) for r, e in enumerate(seen) if e is not None]
        if bad:
            raise RuntimeError("MaD> rank %d failed (%s); stopping all %d ranks" % (bad[0][0], bad[0][1], len(seen)))
        return out

    # ------------------------------------------------------------------ driver
    def run(self, transform_subunits=False, detect_sigma=2.0, presmooth_sigma=1, ori_eqsp_size=112, dsc_eqsp_size=16,
            dsc_subregions=64, patch_size=16, cc_threshold=0.6, weight_threshold=4, n_samples=60):
        self.transform_subunits = transform_subunits
        if not self.check_preprocess_data():
            return
        self.get_descriptors(detect_sigma=detect_sigma, presmooth_sigma=presmooth_sigma, patch_size=patch_size,
                             ori_eqsp_size=ori_eqsp_size, dsc_eqsp_size=dsc_eqsp_size, dsc_subregions=dsc_subregions)
        self.get_solutions(cc_threshold=cc_threshold, weight_threshold=weight_threshold, n_samples=n_samples)

    def check_preprocess_data(self):
        if self.input_map is None or not (len(self.input_subunits) + len(self.input_ensembles)):
            print("MaD> Make sure you have defined at least one component and a density map")
            if self.input_map is not None:
                print("     > Map: %s" % self.input_map)
            for k in self.input_subunits:
                print("     > Subunit: %s" % k)
            return False      # the reference falls through and crashes later (MaD.py:104-113)
        rank, world, dist = self._ranks()
        if world == 1:
            self._prep_files_folders()
            return True
        # one results folder for the job: rank 0 prepares it, everybody learns its name and the processed inputs
        state = [None]
        if rank == 0:
            self._all_ranks(dist, self._prep_files_folders)
            state = [(self.out_folder, self.processed_map, getattr(self, "voxsp", None), self.processed_subunits, self.processed_ensembles)]
        else:
            self._all_ranks(dist, lambda: None)
        dist.broadcast_object_list(state, src=0)
        self.out_folder, self.processed_map, self.voxsp, self.processed_subunits, self.processed_ensembles = state[0]
        return True

    def _cache_name(self, key, detect_sigma, presmooth_sigma, patch_size, ori_eqsp_size, dsc_eqsp_size):
        # MaD.py:118 (the trailing 64 is a literal in the reference)
        return (f"dsc_db/{key}_res{self.resolution}_iso{self.isovalue}_detSig{detect_sigma}_presmooth{presmooth_sigma}"
                f"_patch{patch_size}_orieqsp{ori_eqsp_size}_dsceqsp{dsc_eqsp_size}_subregions{64}.h5")

    def get_descriptors(self, detect_sigma=2.0, presmooth_sigma=1, ori_eqsp_size=112, dsc_eqsp_size=16, dsc_subregions=64, patch_size=16):
        def described(key, struct, what):
            name = self._cache_name(key, detect_sigma, presmooth_sigma, patch_size, ori_eqsp_size, dsc_eqsp_size)
            if self._cache_exists(name):
                rows = self._load_descriptors(name)
                print("MaD> %i descriptors for %s found in database" % (len(rows), key))
                return rows, name
            print("\nMaD> Processing %s %s" % (what, key))
            rows = self._describe_struct(struct, detect_sigma, presmooth_sigma, ori_eqsp_size, dsc_eqsp_size, dsc_subregions, patch_size)
            self._save_descriptors(rows, name)
            return rows, name

        rank, world, dist = self._ranks()
        if world > 1:
            # the structures to describe, in the reference's order, dealt round-robin; a rank describes its own (into the cache:
            # dsc_db/ is the hand-over), then everybody loads the map's rows and refers to the others by cache name
            jobs = [(self.map_name, self.processed_map, "map")] + [(k, v[0], "subunit") for k, v in self.processed_subunits.items()]
            for ek in self.processed_ensembles:
                jobs += [(fk, v[0], "frame") for fk, v in self.processed_ensembles[ek].items()]
            names = {}
            for key, struct, what in jobs:
                names[key] = self._cache_name(key, detect_sigma, presmooth_sigma, patch_size, ori_eqsp_size, dsc_eqsp_size)
            self._all_ranks(dist, lambda: [described(key, struct, what) for i, (key, struct, what) in enumerate(jobs) if i % world == rank])
            dist.barrier()
            self.map_dsc = self._load_descriptors(names[self.map_name])
            for k in self.processed_subunits:
                self.dsc_dict[k] = names[k]
            for ek in self.processed_ensembles:
                self.dsc_dict[ek] = {}
                for fk in self.processed_ensembles[ek]:
                    self.dsc_dict[fk] = names[fk]
            return
        self.map_dsc, _ = described(self.map_name, self.processed_map, "map")
        for k in self.processed_subunits:
            self.dsc_dict[k], _ = described(k, self.processed_subunits[k][0], "subunit")
        for ek in self.processed_ensembles:
            self.dsc_dict[ek] = {}
            print("\nMaD> Describing ensemble %s" % ek)
            for fk in self.processed_ensembles[ek]:
                _, name = described(fk, self.processed_ensembles[ek][fk][0], "frame")
                self.dsc_dict[fk] = name      # frames are re-loaded when matched (MaD.py:158-162)

    def get_solutions(self, cc_threshold=0.6, weight_threshold=4, n_samples=120):
        rank, world, dist = self._ranks()
        if world > 1:
            # MaD.py:165-190: one independent docking per subunit / frame -> dealt round-robin (mad_amd.dist.shard_round_robin); the
            # file lists come back in the reference's order on every rank
            from .dist import shard_round_robin
            jobs = [("sub", k, k) for k in self.processed_subunits]
            for ek in self.processed_ensembles:
                jobs += [("frame", ek, fk) for fk in self.processed_ensembles[ek]]
            mine = {}

            def dock_mine():
                for kind, owner_key, key in shard_round_robin(jobs, rank, world):
                    pdbfile, n_copies = self.processed_subunits[key] if kind == "sub" else self.processed_ensembles[owner_key][key]
                    mine[(kind, owner_key, key)] = self._match_filter_refine(pdbfile, n_copies, key, cc_threshold, weight_threshold, n_samples)

            self._all_ranks(dist, dock_mine)
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            files_of = {}
            for part in everyone:
                files_of.update(part)
            for kind, owner_key, key in jobs:
                files = files_of[(kind, owner_key, key)]
                if kind == "sub":
                    if len(files):
                        self.buildable_subunits[key] = [self.processed_subunits[key][1], files]
                else:
                    ensemble = self.processed_ensembles[owner_key]
                    if owner_key not in self.buildable_subunits:
                        self.buildable_subunits[owner_key] = [ensemble[list(ensemble.keys())[0]][1], []]
                    self.buildable_subunits[owner_key][1].extend(files)
            return
        for k in self.processed_subunits:
            pdbfile, n_copies = self.processed_subunits[k]
            files = self._match_filter_refine(pdbfile, n_copies, k, cc_threshold, weight_threshold, n_samples)
            if len(files):
                self.buildable_subunits[k] = [n_copies, files]
        for ek in self.processed_ensembles:
            ensemble = self.processed_ensembles[ek]
            n_copies = ensemble[list(ensemble.keys())[0]][1]
            self.buildable_subunits[ek] = [n_copies, []]
            for fk in ensemble:
                pdbfile, n_copies = ensemble[fk]
                self.buildable_subunits[ek][1].extend(self._match_filter_refine(pdbfile, n_copies, fk, cc_threshold, weight_threshold, n_samples))

    def build_assembly(self, max_models=10, max_overlap_complex=0.1):
        """MaD.py:192-222; the overlap table and the model CCCs are device calls (mad_amd/assembly.py)."""
        from . import assembly
        rank, world, dist = self._ranks()
        if world > 1:      # one rank builds the models (the combinatorics are sequential); the others wait for its files
            out = self._all_ranks(dist, lambda: assembly.build_assembly(self, max_models=max_models, max_overlap_complex=max_overlap_complex) if rank == 0 else None)
            dist.barrier()
            return out
        return assembly.build_assembly(self, max_models=max_models, max_overlap_complex=max_overlap_complex)

    def _build_from_single(self, sub_key, homomultimer=False):
        from . import assembly
        return assembly.build_from_single(self, sub_key, homomultimer=homomultimer)

    def _build_models(self, sub_sol_dict):
        from . import assembly
        return assembly.build_models(self, sub_sol_dict)

    def _write_complex_from_components(self, components, outname):
        from . import assembly
        assembly.write_complex(components, outname)

    def score_ensembles(self):
        if not self.processed_ensembles:
            print("MaD> No ensembles were provided and/or processed")
            return
        if self._ranks()[0] != 0:      # the tables are printed once
            return
        import csv
        for ek, ensemble in self.processed_ensembles.items():
            ranking = []
            for fk in sorted(ensemble):
                path = os.path.join(self.out_folder, "Solutions_refined_%s.csv" % fk)
                if not os.path.exists(path):
                    continue
                with open(path) as fh:
                    rows = list(csv.DictReader(fh))
                if rows:
                    ranking.append([fk] + [float(np.mean([float(r[c]) for r in rows])) for c in ("Repeatability", "Weight", "mCC", "RWmCC")])
            print("MaD> Ranking for ensemble %s: " % ek)
            for col, title in ((1, "Repeatability"), (2, "Weight"), (3, "Cross-corr."), (4, "MaD score")):
                print("%s     Top 3 - %s:" % ("" if col == 1 else "\n", title))      # blank line between the tables (MaD.py:267-273)
                for i, r in enumerate(sorted(ranking, key=itemgetter(col), reverse=True)[:3]):
                    print("     %i: %6.2f %s" % (i + 1, r[col], r[0]))

    # ------------------------------------------------------------------ files and folders
    def _prep_files_folders(self):
        for d in ("results", "dsc_db"):
            if not os.path.exists(d):
                os.mkdir(d)
        subs = ["%sx%i" % (k, self.input_subunits[k][1]) for k in sorted(self.input_subunits)]
        ens = ["%sx%i" % (k, self.input_ensembles[k][list(self.input_ensembles[k].keys())[0]][1]) for k in sorted(self.input_ensembles)]
        out = f"results/{self.map_name}_{'.'.join(subs + ens)}_res{self.resolution:.3f}_iso{self.isovalue:.3f}"
        if os.path.exists(out):
            i = 1
            while os.path.exists("%s_%i" % (out, i)):
                i += 1
            out = "%s_%i" % (out, i)
        os.mkdir(out)
        self.out_folder = out
        print("MaD> Created output folder: %s" % out)
        init_path = os.path.join(out, "initial_files")
        os.mkdir(init_path)
        ext = os.path.splitext(self.input_map)[-1].lower()
        if ext in (".sit", ".situs", ".mrc", ".map"):
            lo_map = Dmap(self.input_map, isovalue=self.isovalue)
            lo_map.reduce_void()
            self.voxsp = lo_map.voxsp
            self.processed_map = os.path.join(init_path, "%s_mad.mrc" % self.map_name)
            lo_map.write_to_mrc(self.processed_map)
        elif ext == ".pdb":
            print("MaD> PDB provided for density map: %s" % self.input_map)
            print("     Simulating at specified resolution and voxel spacing of 1.2 angstroms")
            self.voxsp = 1.2
            self.processed_map = os.path.join(init_path, "%s_simulated_map.mrc" % self.map_name)
            PDB(self.input_map).structure_to_density(self.resolution, self.voxsp, outname=self.processed_map)
        else:
            print("MaD> ERROR: density map not understood (either sit/mrc/map format or PDB for simulated density): %s" % self.input_map)
        for k, (pdb_file, n_copies) in self.input_subunits.items():
            out_name = os.path.join(init_path, "%s.pdb" % k)
            move_copy_structure(pdb_file, out_name, transform=self.transform_subunits)
            self.processed_subunits[k] = [out_name, n_copies]
        for ek, ensemble in self.input_ensembles.items():
            self.processed_ensembles[ek] = {}
            for fk, (pdb_file, n_copies) in ensemble.items():
                out_name = os.path.join(init_path, os.path.split(pdb_file)[-1])
                move_copy_structure(pdb_file, out_name, transform=self.transform_subunits)
                self.processed_ensembles[ek][fk] = [out_name, n_copies]

    # ------------------------------------------------------------------ hot path
    def _describe_struct(self, struct, detect_sigma, presmooth_sigma, ori_eqsp_size, dsc_eqsp_size, dsc_subregions, patch_size):
        """MaD.py:358-368.  As in the reference, only patch_size reaches the stages."""
        ms = MapSpace(struct, resolution=self.resolution, voxelsp=self.voxsp, sig_init=detect_sigma, sig_presmooth=presmooth_sigma)
        det = Detector()
        ori = Orientator(ori_radius=patch_size)
        dsc = Descriptor(dsc_radius=patch_size)
        ms.build_space()
        anchors = det.find_anchors(ms)
        oriented = ori.assign_orientations(ms, anchors)
        rows = dsc.generate_descriptors(ms, oriented)
        ms.release_device()
        return rows

    def _rowset(self, lib, dsc_list):
        key = id(dsc_list)
        hit = self._rowsets.get(key)
        if hit is None or hit[0] is not dsc_list or hit[1].dev.lib is not lib:
            hit = (dsc_list, _RowSet(lib, dsc_list))
            self._rowsets[key] = hit
        return hit[1]

    def _match_device(self, lo_dsc_list, hi_dsc_list, anchor_dist_thresh, cc_threshold, k):
        lib = _lib.get_lib()
        lo, hi = self._rowset(lib, lo_dsc_list), self._rowset(lib, hi_dsc_list)
        top, idx, stats = lib.match_topk(hi.dev, lo.dev, cc_threshold, anchor_dist_thresh, k)
        if stats["n_pairs"]:
            uh, ul = lib.match_used(len(hi.anchors), len(lo.anchors))
            hi_cloud, lo_cloud = hi.anchors[uh], lo.anchors[ul]
        else:
            hi_cloud, lo_cloud = np.zeros((0, 3)), np.zeros((0, 3))
        return lib, lo, hi, top, idx, stats, lo_cloud, hi_cloud

    def _match_dsc(self, lo_dsc_list, hi_dsc_list, anchor_dist_thresh=4, cc_threshold=0.65):
        """MaD.py:414-453: (list of float64[23] rows in row-major pair order, lo_mapcoords, hi_mapcoords)."""
        lib, lo, hi, _, _, stats, lo_cloud, hi_cloud = self._match_device(lo_dsc_list, hi_dsc_list, anchor_dist_thresh, cc_threshold, 1)
        res = lib.match_results(hi.dev, lo.dev, stats["n_pairs"]) if stats["n_pairs"] else np.zeros((0, 23))
        return list(res), lo_cloud, hi_cloud

    def _match_dsc_topk(self, lo_dsc_list, hi_dsc_list, k, anchor_dist_thresh=4, cc_threshold=0.65):
        """The first k rows of sorted(_match_dsc(...)[0], key=repeatability, reverse=True) (MaD.py:480),
        selected and ordered on the device, plus the two clouds."""
        _, _, _, top, _, _, lo_cloud, hi_cloud = self._match_device(lo_dsc_list, hi_dsc_list, anchor_dist_thresh, cc_threshold, k)
        return top, lo_cloud, hi_cloud

    def _match_filter_refine(self, pdbfile, n_copies, k, cc_threshold, weight_threshold, n_samples):
        n_samples_sub = int(n_samples * n_copies)
        print("MaD> Matching descriptors (%s vs. %s) (cc = %.2f)..." % (self.map_name, k, cc_threshold))
        from_cache = isinstance(self.dsc_dict[k], str)
        hi_list = self._load_descriptors(self.dsc_dict[k]) if from_cache else self.dsc_dict[k]
        top, map_anchors, comp_anchors = self._match_dsc_topk(self.map_dsc, hi_list, n_samples_sub, cc_threshold=cc_threshold)
        if from_cache:      # an ensemble frame re-loaded for this match (MaD.py:158-162, 379-380): one frame resident at a time
            hit = self._rowsets.pop(id(hi_list), None)
            if hit is not None:
                hit[1].dev.close()
        if not len(top):
            print("MaD> No descriptor pair above the threshold for %s" % k)
            return []
        print("MaD> Filtering descriptor pairs (map %s vs. structure %s) (weight=%i, n_samples=%i*%i)..." % (self.map_name, k, weight_threshold, n_samples, n_copies))
        filtered = self._filter_dsc_pairs(pdbfile, top, map_anchors, comp_anchors, wthresh=weight_threshold, n_samples=n_samples_sub, presorted=True)
        print("MaD> Refining %s in %s..." % (self.map_name, k))
        refined = self._refine_filtered_solutions(pdbfile, filtered, map_anchors, comp_anchors)
        return self._save_solutions_refined(refined, k)

    def _filter_dsc_pairs(self, pdbfile, match_data, lo_cloud, hi_cloud, wthresh=4, n_samples=200, presorted=False):
        """Greedy clustering of the best poses by cloud RMSD (MaD.py:456-553).  Host, float64."""
        rmsdcloud_thresh = 10
        data = np.array(match_data if presorted else sorted(match_data, key=itemgetter(REPEAT), reverse=True))
        chain = PDB(pdbfile)
        chain_init = chain.get_coords().copy()
        init_hi_cloud = hi_cloud.copy()

        def moved_cloud(s):
            return np.dot(init_hi_cloud - s[HI_COORD], np.array([s[R1], s[R2], s[R3]]).T) + s[LO_COORD]

        best = data[0]
        cand_ids = [0]
        cand_clouds = [moved_cloud(best)]
        weights = {0: 1}
        members = {0: [[best[HI_COORD], best[LO_COORD], best[HI_BIN], best[LO_BIN]]]}
        counter = 1
        for s in data[1:n_samples]:
            cur = moved_cloud(s)
            rmsd = np.sqrt(np.sum(np.square(cand_clouds - cur), axis=(1, 2)) / len(cur))
            if np.amin(rmsd) > rmsdcloud_thresh:
                cand_ids.append(counter)
                cand_clouds.append(cur)
                weights[counter] = 1
                members[counter] = [[s[HI_COORD], s[LO_COORD], s[HI_BIN], s[LO_BIN]]]
            else:
                owner = cand_ids[int(np.argmin(rmsd))]
                weights[owner] += 1
                members[owner].append([s[HI_COORD], s[LO_COORD], s[HI_BIN], s[LO_BIN]])
            counter += 1
        rep_thresh = max(5, best[REPEAT] * 0.3)
        out = []
        for cand in cand_ids:
            s = data[cand]
            w = weights[cand]
            if w < wthresh or s[REPEAT] < rep_thresh:
                continue
            Rt = np.array([s[R1], s[R2], s[R3]]).T
            chain.set_coords(chain_init)
            chain.translate_atoms(-s[HI_COORD])
            chain.rotate_atoms(Rt)
            chain.translate_atoms(s[LO_COORD])
            out.append([s[HI_COORD], s[LO_COORD], Rt, s[D_CC], w, s[REPEAT], s[REPEAT] * w, deepcopy(chain), members[cand]])
        return sorted(out, key=itemgetter(6), reverse=True)

    def _refine_filtered_solutions(self, pdbfile, filtered_candidate_list, lo_cloud, hi_cloud):
        """MaD.py:556-629 with all candidates refined in ONE launch (one workgroup each)."""
        from scipy.spatial import cKDTree
        hi_pdb = PDB(pdbfile)
        hi_init = hi_pdb.get_coords().copy()
        dmap = Dmap(self.processed_map)
        if not len(filtered_candidate_list):
            return []
        # placement, refinement, density simulation and CCC of every candidate in ONE device call (MaD.py:566-575, 613-616): the poses go
        # in, the refined coordinates and one score per candidate come back
        n_c = len(filtered_candidate_list)
        coords, _, _, ccc_all = dock_refine_score_many(dmap, hi_init, hi_pdb.atom_masses(), np.array([c[0] for c in filtered_candidate_list]),
                                                       np.array([c[1] for c in filtered_candidate_list]),
                                                       np.array([c[2] for c in filtered_candidate_list]).reshape(n_c, 9), self.resolution,
                                                       n_steps=500, max_step_size=1, min_step_size=0.1)
        tree = cKDTree(lo_cloud)
        refined = []
        for cand, xyz, ccc in zip(filtered_candidate_list, coords, ccc_all):
            weight, clustered = cand[4], cand[8]
            if np.any(np.isnan(xyz)):
                continue
            R, T = get_rototrans_SVD(hi_init, xyz)
            moved = np.dot(hi_cloud, R) + T
            dist, _ = tree.query(moved, distance_upper_bound=dmap.voxsp * 1.5)
            repeat = 100 * np.count_nonzero(dist < dmap.voxsp * 2) / hi_cloud.shape[0]
            if repeat > 0:
                sol = deepcopy(hi_pdb)
                sol.set_coords(xyz)
                refined.append([sol, moved[dist < dmap.voxsp * 2], repeat, weight, clustered, float(ccc)])
        final = []
        for sol, corresp, repeat, weight, clustered, ccc in refined:
            if final:
                rmsds = [sol.get_rmsdCA_with(f[0]) for f in final]
                if np.min(rmsds) < 6:      # a clone of an earlier solution: merge (MaD.py:609-612)
                    j = int(np.argmin(rmsds))
                    final[j][3] += weight
                    final[j][5].extend(clustered)
                    continue
            final.append([sol, corresp, repeat, weight, ccc, clustered])
        for sol in final:
            sol.append(sol[2] * sol[3] * sol[4])
        return sorted(final, key=itemgetter(-1), reverse=True)

    # ------------------------------------------------------------------ I/O
    @staticmethod
    def _npz_name(name):
        return name[:-3] + ".npz" if name.endswith(".h5") else name + ".npz"

    def _cache_exists(self, name):
        return (h5py is not None and os.path.exists(name)) or os.path.exists(self._npz_name(name))

    def _save_descriptors(self, df_list, outname):
        data = dict(
            dsc=np.array([df.lin_ar_subeqsp for df in df_list], dtype=np.int16),
            info=np.array([[df.index, df.main_bin, df.sec_bin, df.oct_scale, df.eqsp_size, df.subeqsp_size] for df in df_list]).astype(np.uint16),
            coords=np.array([[df.coords, df.map_coords, df.subv_map_coords] for df in df_list], dtype=np.float64),
            rot=np.array([df.Rfinal for df in df_list], dtype=np.float64))
        if h5py is not None:
            with h5py.File(outname, "w") as hf:
                for k, v in data.items():
                    hf.create_dataset(k, data=v)
        else:
            np.savez(self._npz_name(outname), **data)

    def _load_descriptors(self, input_name):
        if h5py is not None and os.path.exists(input_name):
            with h5py.File(input_name, "r") as hf:
                data = {k: np.array(hf.get(k)) for k in ("dsc", "info", "coords", "rot")}
        else:
            with np.load(self._npz_name(input_name)) as z:
                data = {k: z[k] for k in ("dsc", "info", "coords", "rot")}
        rows = []
        for d, c, i, r in zip(data["dsc"], data["coords"], data["info"], data["rot"]):
            df = DensityFeature()
            df.set_from_file_dsc(int(i[0]), int(i[1]), int(i[2]), int(i[3]), int(i[4]), int(i[5]), c[0], c[1], c[2], r, d)
            rows.append(df)
        return rows

    def _save_solutions_refined(self, refined_solutions, sub_key):
        sol_path = os.path.join(self.out_folder, "individual_solutions")
        anchor_path = os.path.join(sol_path, "anchor_files")
        os.makedirs(anchor_path, exist_ok=True)
        sep = "------------------------------------------"
        print("\n" + sep + "\n|  # | Repeat | Weight |   mCC  |  RWmCC |\n" + sep)
        table, files = [], []
        for idx, (pdb, corresp, repeat, weight, ccc, clustered, score) in enumerate(refined_solutions):
            fname = os.path.join(sol_path, "sol_%s_%i.pdb" % (sub_key, idx))
            pdb.write_pdb(fname)
            files.append(fname)
            self._save_coords_as_pdb(corresp, os.path.join(anchor_path, "corresp_anchors_%s_%i.pdb" % (sub_key, idx)))
            print("| %2i | %6.2f | %6i | %6.2f | %6.2f |" % (idx, repeat, weight, ccc, score))
            table.append([idx, repeat, weight, ccc, score])
        print(sep + "\n")
        if table:
            with open(os.path.join(self.out_folder, "Solutions_refined_%s.csv" % sub_key), "w") as fh:
                fh.write("ID,Repeatability,Weight,mCC,RWmCC\n")
                def num(v):      # as pandas.to_csv prints it: shortest repr in the value's own precision
                    return str(v) if isinstance(v, (np.floating, np.integer)) else repr(v)
                for row in table:
                    fh.write("%i,%s,%s,%s,%s\n" % (row[0], num(row[1]), num(row[2]), num(row[3]), num(row[4])))
        return files

    def _save_coords_as_pdb(self, coords, outname):
        with open(outname, "w") as fh:
            for i, c in enumerate(coords):
                fh.write("%-6s%5i  %-3s %3s%2s%4i    %8.3f%8.3f%8.3f%6.2f%6.2f          %-2s\n"
                         % ("ATOM", i % 100000, "O", "EPC", "E", i % 10000, c[0], c[1], c[2], 1.0, i / len(coords), "O"))      # B = rank in the list (MaD.py:1001)
