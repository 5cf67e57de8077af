"""Assembly building from the refined solutions.

What `MaD.build_assembly` does in the reference (mad/MaD.py:192-222, 632-843): solutions of a subunit present in
several copies are combined into clash-free sub-complexes, sub-complexes of different subunits into assembly models,
each ranked by the pairwise occupancy overlap of low-resolution densities (resolution 5, voxel 2, isovalue 0.2,
MaD.py:669,760) and the final models scored by their CCC with the map.

The numerical part -- n density simulations and the n x n overlap table -- is ONE device call (`mad_overlap_matrix`:
the grids never leave HBM); the CCC of each model is `mad_density_ccc`.  The combinatorics are small host logic, kept
here as pure functions of the overlap table so that they can be checked without a GPU.
"""
import csv
import os
from itertools import combinations, product
from operator import itemgetter

import numpy as np

from . import _lib
from .Dmap import Dmap
from .PDB import PDB

OVERLAP_RES, OVERLAP_VOXSP, OVERLAP_ISO = 5, 2, 0.2      # MaD.py:669, 760
HEADER_CSV = ["#", "CC", "Sum(O)", "Std(O)", "Max(O)", "Composition"]
HEADER = "    # |   CC   | Sum(O) | Std(O) | Max(O) | Composition"


# ---------------------------------------------------------------------------- pure host logic
def rank_copies(overlap, n_copies):
    """Candidate sub-complexes of ONE subunit (MaD.py:656-697): every n_copies-subset of its solutions with
    [indices, sum/n_copies, std, max] of the pairwise overlaps, sorted by the maximum (stable, subsets in
    lexicographic order)."""
    n_sol = len(overlap)
    if n_copies == 1:
        return [[(s,), 0, 0, 0] for s in range(n_sol)]
    out = []
    for subset in combinations(range(n_sol), n_copies):
        vals = [overlap[a, b] for a, b in combinations(subset, 2)]
        out.append([subset, np.sum(vals) / n_copies, np.std(vals), np.max(vals)])
    return sorted(out, key=itemgetter(3))


def rank_models(overlap, groups):
    """Assembly models from one sub-complex per subunit (MaD.py:797-807): `groups` lists, per subunit, the rows of
    the overlap table that belong to it.  The statistics run over the full k x k block of the (upper-triangular)
    table, zeros of the diagonal and the lower triangle included, as in the reference; sorted by the sum."""
    out = []
    for pick in product(*groups):
        pick = np.array(pick)
        block = overlap[np.ix_(pick, pick)].T.ravel()      # meshgrid(c, c) order of MaD.py:800-801
        out.append([pick, np.sum(block), np.std(block), np.max(block)])
    return sorted(out, key=itemgetter(1))


def select_models(ranked, max_models, max_overlap):
    """The prefix of a ranked list the reference writes out (MaD.py:726-728, 826-828): at most max_models, and
    after the first one only while the maximum overlap stays within max_overlap."""
    keep = []
    for cnt, cand in enumerate(ranked):
        if cnt >= max_models or (cand[3] > max_overlap and cnt):
            break
        keep.append(cand)
    return keep


def format_overlap_table(overlap, labels, wide=False):
    """Rows of the table printed at MaD.py:689-697 (wide=False) / 786-795 (wide=True: labels right-aligned to 5 and
    padded to the longest subunit key)."""
    width = len(str(len(overlap)))
    longest = max(len(l.split(".", 1)[1]) for l in labels) if labels else 0
    rows = []
    for idx, vals in enumerate(overlap):
        name = labels[idx]
        if wide:
            row = "%5s%s%s | " % (name, " " * (width - len(str(idx))), " " * (longest - len(name.split(".", 1)[1])))
        else:
            row = "%s%s | " % (name, " " * (width - len(str(idx))))
        row += "".join("   0  " if v == 0.0 else "%.3f " % v for v in vals)
        rows.append(row)
    return rows


def write_complex(components, outname):
    """Concatenate placed components into one PDB, chains relabelled A, B, ... at every atom serial 1, a TER line
    between chains (MaD.py:961-982)."""
    chain = "@"
    with open(outname, "w") as out:
        for comp in components:
            pdb = comp if isinstance(comp, PDB) else PDB(comp)
            for i in range(pdb.n_atoms):
                serial, name, resname, _, resnum, elem, rec = pdb.info[i]
                if serial == 1:
                    chain = chr(ord(chain) + 1)
                    if chain != "A":
                        out.write("TER\n")
                atom = "%-4s" % name if len(name) == 4 else " %-3s" % name
                x, y, z = pdb.coords[i]
                out.write("%-6s%5i %s %3s%2s%4s    %8.3f%8.3f%8.3f%6.2f%6.2f          %-2s\n"
                          % (rec, serial, atom, resname, chain, resnum, x, y, z, 1.0, 0.0, elem))


def write_ranking(path, rows):
    """complex_ranking.csv as pandas writes it at MaD.py:741-742 (the composition is a python list of strings)."""
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow(HEADER_CSV)
        for cnt, ccc, s_sum, s_std, s_max, comp in rows:
            w.writerow([cnt, repr(float(ccc)), _num(s_sum), _num(s_std), _num(s_max), str(list(comp))])


def _num(v):
    return repr(int(v)) if isinstance(v, (int, np.integer)) else repr(float(v))


# ---------------------------------------------------------------------------- device part
def overlap_table(files, lib=None):
    """Pairwise get_overlap of the low-resolution densities of the given PDB files (upper triangle)."""
    lib = lib if lib is not None else _lib.get_lib()
    pdbs = [PDB(f) for f in files]
    return lib.overlap_matrix([p.coords for p in pdbs], [p.atom_masses() for p in pdbs], resolution=OVERLAP_RES,
                              voxsp=OVERLAP_VOXSP, density_isovalue=OVERLAP_ISO)


def _score_and_report(mad, ranked, files, out_dir, max_overlap):
    """Write Model_<n>.pdb for the selected candidates, score them against the map (structure_to_density(4, voxsp)
    + get_CCC_with_grid, MaD.py:733-736) and print / save the ranking."""
    dmap = Dmap(mad.processed_map)
    print("MaD> Final models docked in map %s: " % mad.map_name)
    print()
    print(HEADER)
    print("-" * len(HEADER))
    rows = []
    for cnt, (idx, s_sum, s_std, s_max) in enumerate(select_models(ranked, mad.max_models, max_overlap), start=1):
        outname = os.path.join(out_dir, "Model_%i.pdb" % cnt)
        write_complex([files[i] for i in idx], outname)
        model = PDB(outname)
        grid, x0, y0, z0 = model.structure_to_density(4, dmap.voxsp)
        ccc = dmap.get_CCC_with_grid(grid, x0, y0, z0)
        comp = [str(i) for i in idx]
        print("  %3i | %6.2f  %6.2f   %6.2f   %6.2f  | %s" % (cnt, ccc, s_sum, s_std, s_max, ".".join(comp)))
        rows.append([cnt, ccc, s_sum, s_std, s_max, comp])
    print("-" * len(HEADER))
    if rows:
        write_ranking(os.path.join(mad.out_folder, "complex_ranking.csv"), rows)
    return rows


def build_from_single(mad, sub_key, homomultimer=False):
    """MaD._build_from_single (MaD.py:632-742)."""
    out_dir = os.path.join(mad.out_folder, "assembly_models" if homomultimer else "subcomplexes")
    os.makedirs(out_dir, exist_ok=True)
    n_copies, solutions = mad.buildable_subunits[sub_key]
    if n_copies > len(solutions):
        print("MaD> Not enough solutions to cover all copies for subunit %s !" % sub_key)
        print("     Maybe try increasing n_samples or reducing min_cc/wthresh ?")
        print("     In the meantime, trying with available solutions...")
        n_copies = len(solutions)
    if n_copies == 1:
        overlap = np.zeros((len(solutions), len(solutions)))
    else:
        overlap = overlap_table(solutions)
        print("MaD> Pairwise overlaps between solutions of %s:" % sub_key)
        print()
        for row in format_overlap_table(overlap, ["%i.%s" % (i, sub_key) for i in range(len(solutions))]):
            print(row)
        print()
        print("MaD> Assembling %i copies of chain %s from %i solutions..." % (n_copies, sub_key, len(solutions)))
    ranked = rank_copies(overlap, n_copies)
    if homomultimer:
        # the reference tests the literal 0.1 here, not max_overlap_complex (MaD.py:727)
        return _score_and_report(mad, ranked, solutions, out_dir, 0.1)
    written = []
    for s_idx, (idx, _, _, s_max) in enumerate(ranked):
        if s_max > mad.max_overlap_complex:
            continue
        code = "_".join("%s%i" % (sub_key, x) for x in idx)
        outname = os.path.join(out_dir, "SubComplex%s_%i_%s.pdb" % (sub_key, s_idx, code))
        write_complex([solutions[i] for i in idx], outname)
        written.append(outname)
    if n_copies > 1:
        print("MaD> Generated %i subcomplexes from component %s" % (len(written), sub_key))
    return written


def build_models(mad, sub_sol_dict):
    """MaD._build_models (MaD.py:745-843)."""
    print("MaD> Building assembly models from %i components..." % len(sub_sol_dict))
    files, labels, groups = [], [], []
    for sub_key, sols in sub_sol_dict.items():
        groups.append(list(range(len(files), len(files) + len(sols))))
        for sol in sols:
            labels.append("%i.%s" % (len(files), sub_key))
            files.append(sol)
    overlap = overlap_table(files)
    print("MaD> Pairwise overlaps between solutions of %s:" % (list(sub_sol_dict)[-1] if sub_sol_dict else ""))
    print()
    for row in format_overlap_table(overlap, labels, wide=True):
        print(row)
    print()
    ranked = rank_models(overlap, groups)
    out_dir = os.path.join(mad.out_folder, "assembly_models")
    os.makedirs(out_dir, exist_ok=True)
    return _score_and_report(mad, ranked, files, out_dir, mad.max_overlap_complex)


def build_assembly(mad, max_models=10, max_overlap_complex=0.1):
    """MaD.build_assembly (MaD.py:192-222)."""
    mad.max_overlap_complex = max_overlap_complex
    mad.max_models = max_models
    if not mad.buildable_subunits:
        print("MaD> No solutions found. Please run() first or adjust parameters if you did not get any solution.")
        return
    if sum(v[0] for v in mad.buildable_subunits.values()) == 1:
        print("MaD> No assembly to build from a monomeric structure")
        return
    if len(mad.buildable_subunits) == 1:
        return build_from_single(mad, next(iter(mad.buildable_subunits)), homomultimer=True)
    subs = {k: build_from_single(mad, k) for k in mad.buildable_subunits}
    return build_models(mad, subs)
