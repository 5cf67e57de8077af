"""Voxel-grid object of the hot path.

Field contract of the reference's `Dmap` (mad/Dmap.py:6-71): `grid3d` float32 [x,y,z],
origin `xi, yi, zi` (Angstrom), box `xb, yb, zb`, `voxsp`, `map_name`, `name`.
`get_CCC_with_grid` (Dmap.py:153-258) runs on the GPU through `mad_ccc`.
`mask_with`, `get_CCC_with_dmap` and the per-voxel text writer are not used by
`MaD.run` / `build_assembly` and are out of scope.
"""
import os
import sys

import numpy as np

from . import _lib, mapio


class Dmap(object):
    def __init__(self, map_name, isovalue=0.0, normalize=True, pad=0):
        if not os.path.isfile(map_name):
            print("Dmap> ERROR: file %s not found" % map_name)
            sys.exit(1)
        ext = os.path.splitext(map_name)[-1].lower()
        if ext in (".sit", ".situs"):
            self.grid3d, self.voxsp, (self.xi, self.yi, self.zi) = mapio.read_situs(map_name, np.float32)
            self.xb, self.yb, self.zb = self.grid3d.shape
        elif ext in (".map", ".mrc"):
            self.grid3d, self.voxsp, (self.xi, self.yi, self.zi), (self.xb, self.yb, self.zb) = mapio.load_mrc_as_xyz(map_name)
        else:
            print("Dmap> ERROR: incompatible extension for map %s" % map_name)
            return
        # threshold (Dmap.py:50-54), optional padding, normalisation to max = 1
        if np.amax(self.grid3d > isovalue):
            self.grid3d[self.grid3d < isovalue] = 0
        else:
            print("Dmap> WARNING: asked isovalue is larger than maximum density found in file (%f). Considering isovalue=0" % np.amax(self.grid3d))
            self.grid3d[self.grid3d < 0] = 0
        if pad:
            self.pad_grid(pad)
        if np.isclose(np.amax(self.grid3d), 0):
            print("Dmap> WARNING: Max value in map is 0")
        if normalize:
            self.grid3d = self.grid3d / np.amax(self.grid3d)
        self.map_name = map_name
        self.name = map_name.split('/')[-1].split('.')[0]

    def reduce_void(self, zeros_padding=10):
        """Crop to the bounding box of the non-zero voxels, then re-pad (Dmap.py:73-90)."""
        nz = np.nonzero(self.grid3d)
        lo = [int(np.amin(a)) for a in nz]
        hi = [int(np.amax(a)) for a in nz]
        self.xi += lo[0] * self.voxsp
        self.yi += lo[1] * self.voxsp
        self.zi += lo[2] * self.voxsp
        self.grid3d = self.grid3d[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
        self.xb, self.yb, self.zb = self.grid3d.shape
        self.pad_grid(zeros_padding)

    def pad_grid(self, pad):
        self.grid3d = np.pad(self.grid3d, pad, mode="constant")
        self.xi -= pad * self.voxsp
        self.yi -= pad * self.voxsp
        self.zi -= pad * self.voxsp
        self.xb, self.yb, self.zb = self.grid3d.shape

    def get_CCC_with_grid(self, grid2, xi2, yi2, zi2, isovalue=0):
        """Un-centred normalised cross-correlation over the overlap box (Dmap.py:153-258).

        Like the reference, voxels below `isovalue` are zeroed in place in BOTH grids."""
        g1 = self.grid3d
        if g1.dtype != np.float32 or not g1.flags.c_contiguous or not g1.flags.writeable:
            g1 = np.ascontiguousarray(g1, dtype=np.float32)
            self.grid3d = g1
        g2 = grid2
        if g2.dtype != np.float32 or not g2.flags.c_contiguous or not g2.flags.writeable:
            g2 = np.ascontiguousarray(grid2, dtype=np.float32)
        ccc = _lib.get_lib().ccc(g1, (self.xi, self.yi, self.zi), g2, (xi2, yi2, zi2), self.voxsp, isovalue)
        if g2 is not grid2:
            try:
                grid2[...] = g2
            except Exception:
                pass
        return ccc

    def write_to_mrc(self, outname):
        mapio.write_mrc(outname, self.grid3d, (self.xi, self.yi, self.zi), self.voxsp)

    def write_to_sit(self, outname):
        print(">Dmap> Writing density map as %s" % outname)      # Dmap.py:379
        mapio.write_situs(outname, self.grid3d, (self.xi, self.yi, self.zi), self.voxsp)
