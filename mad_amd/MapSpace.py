"""Scale-space preparation: the producer of the hot path's input fields.

Mirror of the reference's `MapSpace` (mad/MapSpace.py:12-214): load or simulate the density grid, pad it by
9 voxels, build the 2x cubic-spline upsampled octave (pre-smoothed with sigma 1), the scale-normalised LoG
volumes the detector searches and the Gaussian(sigma)-smoothed gradient fields that orientation and
description sample.  Attributes kept: `grad_list, rgi_space, map_space, gauss_list, grid_list,
voxelsp_list, xi, yi, zi, name`.

SURVEY.md section 8(f) rank 2.  The numerical body runs on the device (`mad_space_build`,
mad_amd/csrc/mad_space.hip): the volumes stay in HBM, the gradient texels are written straight into the
field slots the orientation / descriptor kernels sample (`device_slots`), and the peak search of the detector
reads the LoG volumes where they are (`space`).  The list attributes of the reference are materialised on the
host only when somebody reads them.  There is no CPU fallback; the scipy restatement used to check this
module lives in oracle/scale_space.py (test infrastructure).
"""
import os
import sys

import numpy as np

from . import _lib, mapio
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
        self.space = None
        self._cache = {}

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

    def build_from_grid(self, grid, xi, yi, zi, lib=None):
        """Everything after the file has become a grid (MapSpace.py:116-189), on the device."""
        lib = lib if lib is not None else _lib.get_lib()
        self.release_device()
        if self.map_padding:
            xi -= self.map_padding * self.voxelsp
            yi -= self.map_padding * self.voxelsp
            zi -= self.map_padding * self.voxelsp
        self.xi, self.yi, self.zi = xi, yi, zi
        grid = np.asarray(grid)
        if grid.dtype not in (np.float32, np.float64):
            grid = grid.astype(np.float32)
        slot_up = lib.new_slot() if self.oct_mode in ("up", "both") else -1
        slot_base = lib.new_slot() if self.oct_mode in ("base", "both") else -1
        self.space = _lib.DeviceSpace(lib).build(grid, pad=self.map_padding, oct_mode=self.oct_mode, sig_init=self.sig_init,
                                                 sig_presmooth=self.sig_presmooth, slot_up=slot_up, slot_base=slot_base)
        # list index == DensityFeature.oct_scale when oct_mode is "both": 0 = upsampled, 1 = base
        slots = [slot_up if k == 0 else slot_base for k in self.space.kinds]
        while len(slots) < 2:
            slots.append(-1)
        self._slots = (lib, slots)
        self.voxelsp_list = [self.voxelsp / 2 if k == 0 else self.voxelsp for k in self.space.kinds]
        self._cache = {}

    # -- the reference's list attributes, fetched from the device on first use ----------------------------
    def _volumes(self, what, key):
        if key not in self._cache:
            self._cache[key] = [self.space.download(o, what) for o in range(len(self.space.shapes))]
        return self._cache[key]

    @property
    def grid_list(self):
        return self._volumes(_lib.DeviceSpace.GRID, "grid")

    @property
    def map_space(self):
        return self._volumes(_lib.DeviceSpace.LOG, "log")

    @property
    def gauss_list(self):
        return self._volumes(_lib.DeviceSpace.GAUSS, "gauss")

    @property
    def grad_list(self):
        if "grad" not in self._cache:      # MapSpace.py:187 on the downloaded smoothed volumes
            self._cache["grad"] = [np.moveaxis(np.array(np.gradient(g)), 0, -1) for g in self.gauss_list]
        return self._cache["grad"]

    @property
    def rgi_space(self):
        if "rgi" not in self._cache:
            self._cache["rgi"] = [NearestGradient(g) for g in self.grad_list]
        return self._cache["rgi"]

    # -- device residency ---------------------------------------------------------------
    def device_slots(self, lib):
        """[field slot of list entry 0, of list entry 1] (-1 if absent), filled by build_from_grid."""
        if self._slots is None or self._slots[0] is not lib:
            raise _lib.MadBackendError("MaD> this MapSpace was not built on the requested device context")
        return self._slots[1]

    def release_device(self):
        if self._slots is not None:
            lib, slots = self._slots
            for s in slots:
                if s >= 0 and lib.ctx:
                    lib.free_field(s)
        self._slots = None
        if getattr(self, "space", None) is not None:
            self.space.close()
        self.space = None
