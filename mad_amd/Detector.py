"""Anchor detection (upstream of the hot path).

Mirror of the reference's `Detector` (mad/Detector.py:18-128): anchors are local maxima of the LoG volumes,
refined to sub-voxel precision with a quadratic fit and rejected when the fit wanders or the Hessian has a
positive eigenvalue.

The dense half -- the 3x3x3 local-maximum mask over the whole LoG volume (Detector.py:29) -- runs on the
device where the volume already is (`mad_space_peaks`); only the peak list (a few thousand voxels) and their
13^3 neighbourhoods (`mad_space_patches`) come back, and `check_localize` runs on those with the reference's
own numpy expressions.

PARITY UNPINNED for the peak search: the reference calls `skimage.feature.peak_local_max(grid,
exclude_border=12, threshold_abs=5e-2)` (scikit-image 0.17.2, requirements.txt:5), which is not vendored and
not installed here.  The device kernel follows that version's published behaviour (3x3x3 maximum filter with
zero extension, strict threshold, border exclusion, peaks ordered by descending intensity).  Everything from
`check_localize` on is pinned by fixtures.
"""
import os

import numpy as np

from .DensityFeature import DensityFeature

WALK = 6      # check_localize moves at most 5 voxels per axis and reads one voxel further


class PatchGrid(object):
    """A LoG volume seen through the neighbourhood of one peak: indexable like `grid[x, y, z]` with the
    volume's own coordinates, which is all `check_localize` needs."""

    def __init__(self, patch, centre, shape):
        self.patch, self.shape = patch, shape
        self.off = (int(centre[0]) - WALK, int(centre[1]) - WALK, int(centre[2]) - WALK)

    def __getitem__(self, xyz):
        x, y, z = xyz
        return self.patch[x - self.off[0], y - self.off[1], z - self.off[2]]


class Detector(object):
    def __init__(self):
        self.lowdensity = 0
        self.lowcontrast = 0
        self.saddlepoint = 0
        self.lowratio = 0
        self.largeoffset = 0
        self.badhessian = 0

    def find_anchors(self, ms, outname=""):
        print("MaD> Finding anchors in %s... " % ms.name)
        df_list = []
        for o in range(len(ms.space.shapes)):
            peaks, vals = ms.space.peaks(o, threshold=5e-2, border=12)
            patches = ms.space.patches(o, peaks, WALK)
            vs = ms.voxelsp_list[o]
            for peak, val, patch in zip(peaks, vals, patches):
                ok, coord, subcoord = self.check_localize(PatchGrid(patch, peak, ms.space.shapes[o]), peak)
                if not ok:
                    continue
                df = DensityFeature()
                df.set_detector_info(len(df_list), o, [coord[0], coord[1], coord[2]],
                                     self.get_coord_in_ref_map(coord[0], coord[1], coord[2], ms.xi, ms.yi, ms.zi, vs),
                                     self.get_coord_in_ref_map(subcoord[0], subcoord[1], subcoord[2], ms.xi, ms.yi, ms.zi, vs),
                                     patch.dtype.type(val))
                df_list.append(df)
        if outname and os.path.exists(os.path.split(outname)[0]):
            self.write_df_to_pdb(df_list, outname + ".pdb")
        return df_list

    def check_localize(self, grid, oricoord):
        """Quadratic sub-voxel localisation with saddle rejection (Detector.py:53-123)."""
        # numpy integers, as the peak search hands them over: int64 + float32 offset is a float64 sum (Detector.py:117-119)
        x, y, z = (np.int64(v) for v in oricoord)
        max_off = 0.6
        offset = np.zeros(3)
        H = np.zeros((3, 3))
        found = False
        for _ in range(5):
            c = grid[x, y, z]
            xx = grid[x - 1, y, z] + grid[x + 1, y, z] - 2 * c
            yy = grid[x, y - 1, z] + grid[x, y + 1, z] - 2 * c
            zz = grid[x, y, z - 1] + grid[x, y, z + 1] - 2 * c
            xy = 0.25 * ((grid[x + 1, y + 1, z] - grid[x + 1, y - 1, z]) - (grid[x - 1, y + 1, z] - grid[x - 1, y - 1, z]))
            xz = 0.25 * ((grid[x + 1, y, z + 1] - grid[x + 1, y, z - 1]) - (grid[x - 1, y, z + 1] - grid[x - 1, y, z - 1]))
            yz = 0.25 * ((grid[x, y + 1, z + 1] - grid[x, y + 1, z - 1]) - (grid[x, y - 1, z + 1] - grid[x, y - 1, z - 1]))
            H = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
            G = np.array([0.5 * (grid[x + 1, y, z] - grid[x - 1, y, z]),
                          0.5 * (grid[x, y + 1, z] - grid[x, y - 1, z]),
                          0.5 * (grid[x, y, z + 1] - grid[x, y, z - 1])])
            try:
                offset = -np.dot(np.linalg.inv(H), G)
            except Exception:
                return False, oricoord, oricoord
            if np.all(np.abs(offset) < max_off):
                found = True
                break
            ox, oy, oz = offset
            if ox < -max_off and x - 1 > 0:
                x -= 1
            elif ox > max_off and x + 1 < grid.shape[0] - 1:
                x += 1
            if oy < -max_off and y - 1 > 0:
                y -= 1
            elif oy > max_off and y + 1 < grid.shape[1] - 1:
                y += 1
            if oz < -max_off and z - 1 > 0:
                z -= 1
            elif oz > max_off and z + 1 < grid.shape[2] - 1:
                z += 1
        if not found:
            return False, oricoord, oricoord
        if np.any(np.linalg.eigvals(H) > 0):      # a maximum has no positive curvature
            return False, oricoord, oricoord
        return True, [x, y, z], [x + offset[0], y + offset[1], z + offset[2]]

    def get_coord_in_ref_map(self, x, y, z, xi, yi, zi, voxsp):
        return np.array([x * voxsp + xi, y * voxsp + yi, z * voxsp + zi])

    def write_df_to_pdb(self, df_list, outname):
        with open(outname, "w") as f:
            for i, df in enumerate(df_list):
                c = df.subv_map_coords
                f.write("%-6s%5i  %-3s %3s%2s%4i    %8.3f%8.3f%8.3f%6.2f%6.2f          %-2s\n"
                        % ("ATOM", i % 100000, "CA", "ANC", "A", i % 10000, c[0], c[1], c[2], 1.0, 0.0, "C"))
