"""Rigid-body refinement and the structure helpers around it.

`refine_pdb` has the reference's signature and return value
(mad/structure_utils.py:58-161) and, like it, moves `pdb.coords` in place; the <= 500
dependent gradient-ascent steps run in one persistent HIP workgroup (`k_refine`,
through `mad_refine`).  `refine_many` refines several placements of one structure in a
single launch (one workgroup each).  `move_structure` / `move_copy_structure` are plain
bookkeeping; `get_overlap` (assembly building) counts on the
device through `mad_grid_overlap`.
"""
import zlib

import numpy as np

from . import _lib
from .math_utils import euler_rod_mat
from .PDB import PDB

# What the device holds as "the map candidates are refined in": the context and the grid OBJECT (strong references, so neither
# id can be recycled for another map while this entry lives), its placement, and a fingerprint of its contents.
_uploaded = {"lib": None, "grid": None, "where": None, "print": None}


def _fingerprint(grid):
    """Cheap token of a grid's contents: shape, dtype and the CRC of a strided sample of ~64k voxels (thresholding,
    normalisation, masking and re-loading all change it; a single-voxel edit between two calls may not --
    `invalidate_density()` is there for that)."""
    flat = grid.reshape(-1)
    step = max(1, flat.size // 65536)
    return (grid.shape, str(grid.dtype), zlib.crc32(np.ascontiguousarray(flat[::step]).tobytes()))


def invalidate_density():
    """Forget which map is on the device: the next refinement / CCC call uploads again."""
    _uploaded.update(lib=None, grid=None, where=None, print=None)


def _ensure_density(lib, dmap):
    where = (float(dmap.xi), float(dmap.yi), float(dmap.zi), float(dmap.voxsp))
    fp = _fingerprint(dmap.grid3d)
    if _uploaded["lib"] is lib and lib.ctx and _uploaded["grid"] is dmap.grid3d and _uploaded["where"] == where and _uploaded["print"] == fp:
        return
    lib.upload_density(dmap.grid3d, (dmap.xi, dmap.yi, dmap.zi), dmap.voxsp)
    _uploaded.update(lib=lib, grid=dmap.grid3d, where=where, print=fp)


def _rmsd_before_after(pdb, before, after):
    if len(pdb.CA_idx):
        d = np.square(after[pdb.CA_idx, :] - before[pdb.CA_idx, :])
    else:
        d = np.square(after - before)
    return np.sqrt(np.sum(d, axis=(0, 1)) / d.shape[0])


def refine_pdb(dmap, pdb, n_steps=500, max_step_size=0.5, min_step_size=0.01, idx=-1):
    """-> (CA-RMSD before/after, converged, last step); pdb.coords is updated in place."""
    lib = _lib.get_lib()
    _ensure_density(lib, dmap)
    start = pdb.coords.copy()
    coords, converged, step = lib.refine(start, n_steps=n_steps, max_step=max_step_size, min_step=min_step_size)
    if np.any(np.isnan(coords)):
        pdb.set_coords(coords)
        return np.nan, False, step
    pdb.set_coords(coords)
    return _rmsd_before_after(pdb, start, coords), converged, step


def refine_many(dmap, start_coords, n_steps=500, max_step_size=0.5, min_step_size=0.01):
    """start_coords (n_cand, n_atoms, 3) -> (coords, converged[n_cand], last_step[n_cand])."""
    lib = _lib.get_lib()
    _ensure_density(lib, dmap)
    return lib.refine(np.asarray(start_coords, dtype=np.float64), n_steps=n_steps, max_step=max_step_size, min_step=min_step_size)


def ccc_many(dmap, coords, masses, resolution, isovalue=0):
    """CCC of the simulated densities of several placements of one structure with `dmap`: what
    `pdb.structure_to_density(resolution, dmap.voxsp)` + `dmap.get_CCC_with_grid(grid, x0, y0, z0, isovalue)` give
    one placement at a time (MaD.py:613-616), in one device-resident batch (`mad_density_ccc`).  Like the
    reference's get_CCC_with_grid, the map is left clamped at the isovalue (Dmap.py:160)."""
    lib = _lib.get_lib()
    _ensure_density(lib, dmap)
    out = lib.density_ccc(np.asarray(coords, dtype=np.float64), masses, resolution, 0.0, isovalue)
    if np.any(dmap.grid3d < isovalue):
        dmap.grid3d[dmap.grid3d < isovalue] = 0
        invalidate_density()      # the host copy changed: the device copy is re-made from it next time
    return out


def dock_refine_score_many(dmap, base_coords, masses, hi_coords, lo_coords, rots, resolution, n_steps=500, max_step_size=0.5,
                           min_step_size=0.01, isovalue=0, want_coords=True):
    """MaD._refine_filtered_solutions' device work for all candidate poses of one structure in ONE call (`mad_dock_refine_score`):
    placement (translate by -hi, rotate, translate by lo: MaD.py:566-569), `refine_pdb`, `structure_to_density` and
    `get_CCC_with_grid` (MaD.py:613-616) without the coordinates ever leaving the device, unless asked for.
    -> (coords (n_cand, n, 3) or None, converged, last_step, ccc).  Like get_CCC_with_grid, the map is left clamped at `isovalue`."""
    lib = _lib.get_lib()
    _ensure_density(lib, dmap)
    out = lib.dock_refine_score(base_coords, masses, hi_coords, lo_coords, rots, resolution, n_steps=n_steps, max_step=max_step_size,
                                min_step=min_step_size, density_isovalue=0.0, ccc_isovalue=isovalue, want_coords=want_coords)
    if np.any(dmap.grid3d < isovalue):
        dmap.grid3d[dmap.grid3d < isovalue] = 0
        invalidate_density()
    return out


def move_structure(original_struct, t=None, a=0.375, b=1.735, c=2.452, suffix=""):
    moved = original_struct.replace(".pdb", "_moved%s.pdb" % suffix)
    pdb = PDB(original_struct)
    for axis, ang in (([1, 0, 0], a), ([0, 1, 0], b), ([0, 0, 1], c)):
        pdb.rotate_atoms(euler_rod_mat(axis, ang))
    pdb.translate_atoms(-np.mean(pdb.get_coords(), axis=0) if t is None else t)
    pdb.write_pdb(moved)
    return moved


def move_copy_structure(original_struct, moved_struct, transform=False, t=[150, 0, 0], a=0.375, b=1.735, c=2.452):
    pdb = PDB(original_struct)
    if transform:
        for axis, ang in (([1, 0, 0], a), ([0, 1, 0], b), ([0, 0, 1], c)):
            pdb.rotate_atoms(euler_rod_mat(axis, ang))
        pdb.translate_atoms(-np.mean(pdb.get_coords(), axis=0))
        if len(t):
            pdb.translate_atoms(t)
    pdb.write_pdb(moved_struct)
    return moved_struct


def get_overlap(g1, g2, voxsp, isovalue=1e-8):
    """Fraction of the occupied voxels of grid 1 that grid 2 also occupies -- structure_utils.py:163-259.
    g1, g2 = (grid, x0, y0, z0).  Like the reference, both grids are clamped in place at `isovalue`."""
    def device_view(grid):
        if grid.dtype == np.float32 and grid.flags.c_contiguous and grid.flags.writeable:
            return grid
        return np.ascontiguousarray(grid, dtype=np.float32)
    (grid1, x1, y1, z1), (grid2, x2, y2, z2) = g1, g2
    a, b = device_view(grid1), device_view(grid2)
    common, occupied = _lib.get_lib().grid_overlap(a, (x1, y1, z1), b, (x2, y2, z2), voxsp, isovalue)
    for src, dst in ((a, grid1), (b, grid2)):
        if src is not dst:
            dst[...] = src
    if occupied == 0:
        return 0
    return common / occupied
