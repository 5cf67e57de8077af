"""Scale-space preparation: the producer of the hot path's input fields.

Mirror of the reference's `MapSpace` (mad/MapSpace.py:12-214): load or simulate the
density grid, pad it by 9 voxels, build the 2x cubic-spline upsampled octave (pre-
smoothed with sigma 1), the scale-normalised LoG volumes the detector searches and the
Gaussian(sigma)-smoothed gradient fields `grad_list` that orientation and description
sample.  Attributes kept: `grad_list, rgi_space, map_space, gauss_list, grid_list,
voxelsp_list, xi, yi, zi, name`.

This stage is SURVEY.md section 8(f) rank 2 ("next"): it runs on the host with the same
scipy calls as the reference and its gradient fields are uploaded once per structure
(`device_slots`).  `rgi_space[o]` is kept for API compatibility only (a nearest-
neighbour lookup object); the HIP descriptor kernel gathers from the uploaded field.
"""
import os
import sys

import numpy as np
from scipy.interpolate import interp1d
from scipy.ndimage import gaussian_filter, gaussian_laplace

from . import mapio
from .PDB import PDB


class NearestGradient(object):
    """Callable stand-in for the reference's `RegularGridInterpolator(..., method="nearest")`
    (MapSpace.py:189): same tie rule (fraction <= 0.5 -> lower index) and the same
    ValueError when a point leaves the grid."""

    def __init__(self, values):
        self.values = values

    def __call__(self, pts):
        p = np.asarray(pts, dtype=float)
        shp = p.shape[:-1]
        p = p.reshape(-1, 3)
        idx = []
        for d in range(3):
            n = self.values.shape[d]
            x = p[:, d]
            if not (np.all(x >= 0) and np.all(x <= n - 1)):
                raise ValueError("One of the requested xi is out of bounds in dimension %d" % d)
            i = np.clip(np.floor(x).astype(int), 0, n - 2)
            idx.append(np.where(x - i <= 0.5, i, i + 1))
        return self.values[tuple(idx)].reshape(shp + self.values.shape[3:])


class MapSpace(object):
    def __init__(self, structure_file, resolution=0, voxelsp=0, isovalue=0.0, map_padding=9, oct_mode="both",
                 sig_init=2, sig_presmooth=1):
        self.structure_file = structure_file
        self.isovalue = isovalue
        self.map_padding = map_padding
        self.PDB_mode = False
        self.name = os.path.splitext(os.path.split(structure_file)[-1])[0]
        self.sig_init = sig_init
        self.sig_presmooth = sig_presmooth
        self.oct_mode = oct_mode
        if oct_mode not in ("base", "up", "both"):
            print("MaD> WARNING: #octave not set properly (%s), reverting to 'base'" % oct_mode)
            self.oct_mode = "base"
        self.ext = os.path.splitext(structure_file)[-1].lower()
        if self.ext == ".pdb":
            self.PDB_mode = True
            self.voxelsp = voxelsp
            self.resolution = resolution
            if not voxelsp:
                print("MaD> ERROR: if providing a PDB, voxel spacing is mandatory")
                sys.exit(1)
            if not resolution:
                print("MaD> ERROR: if providing a PDB, resolution is mandatory")
                sys.exit(1)
        elif self.ext not in (".situs", ".sit", ".map", ".mrc"):
            print("MaD> ERROR: please provide a valid structure file (pdb, sit, situs, map or mrc format)")
            print(self.ext)
            sys.exit(1)
        self._slots = None

    def _load_grid(self):
        if self.PDB_mode:
            grid, xi, yi, zi = PDB(self.structure_file).structure_to_density(self.resolution, self.voxelsp, isovalue=self.isovalue)
        elif self.ext in (".situs", ".sit"):
            grid, self.voxelsp, (xi, yi, zi) = mapio.read_situs(self.structure_file, np.float64)
            grid[grid < self.isovalue] = 0
            grid = grid / np.amax(grid).astype(np.float32)      # MapSpace.py:96
        else:
            grid, self.voxelsp, (xi, yi, zi), _ = mapio.load_mrc_as_xyz(self.structure_file)
            grid = grid.astype(np.float32)
            grid[grid < self.isovalue] = 0
        return grid, xi, yi, zi

    def build_space(self):
        print("MaD> Building map space for %s..." % self.name)
        grid, xi, yi, zi = self._load_grid()
        self.build_from_grid(grid, xi, yi, zi)

    def build_from_grid(self, grid, xi, yi, zi):
        """Everything after the file has become a grid (MapSpace.py:116-189)."""
        if self.map_padding:
            grid = np.pad(grid, self.map_padding, mode="constant")
            xi -= self.map_padding * self.voxelsp
            yi -= self.map_padding * self.voxelsp
            zi -= self.map_padding * self.voxelsp
        self.xi, self.yi, self.zi = xi, yi, zi
        xb, yb, zb = grid.shape

        octaves = []
        if self.oct_mode in ("up", "both"):
            # 2x upsampling by successive 1-D cubic splines, then a light pre-smoothing (MapSpace.py:137-146)
            up = grid
            for axis, n in enumerate((xb, yb, zb)):
                up = interp1d(np.arange(0, n, 1), up, axis=axis, kind="cubic")(np.arange(0, n - 0.5, 0.5))
            if self.sig_presmooth:
                up = gaussian_filter(up, sigma=self.sig_presmooth)
            octaves.append((up.astype(np.float32), self.voxelsp / 2))
        if self.oct_mode in ("base", "both"):
            octaves.append((grid, self.voxelsp))
        self.grid_list = [g for g, _ in octaves]
        self.voxelsp_list = [v for _, v in octaves]

        # scale-normalised LoG for the detector (MapSpace.py:169-173)
        self.map_space = []
        for g in self.grid_list:
            log_g = -1 * gaussian_laplace(g, sigma=self.sig_init) * self.sig_init ** 2
            log_g[log_g < 0] = 0.0
            self.map_space.append(log_g)

        # Gaussian-smoothed gradient fields (MapSpace.py:178-189)
        self.gauss_list, self.grad_list, self.rgi_space = [], [], []
        for g in self.grid_list:
            self.gauss_list.append(gaussian_filter(g, self.sig_init))
            self.grad_list.append(np.moveaxis(np.array(np.gradient(self.gauss_list[-1])), 0, -1))
            self.rgi_space.append(NearestGradient(self.grad_list[-1]))
        self._slots = None

    # -- device residency ---------------------------------------------------------------
    def device_slots(self, lib):
        """Upload grad_list once; returns [slot of list entry 0, slot of list entry 1] (-1 if absent).

        List index == DensityFeature.oct_scale (0 = upsampled, 1 = base) when oct_mode is "both"."""
        if self._slots is None or self._slots[0] is not lib:
            slots = []
            for g in self.grad_list:
                s = lib.new_slot()
                lib.upload_field(s, g)
                slots.append(s)
            while len(slots) < 2:
                slots.append(-1)
            self._slots = (lib, slots)
        return self._slots[1]

    def release_device(self):
        if self._slots is not None:
            lib, slots = self._slots
            for s in slots:
                if s >= 0 and lib.ctx:
                    lib.free_field(s)
        self._slots = None
