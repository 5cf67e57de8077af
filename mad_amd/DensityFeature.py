"""Per-anchor record of the hot path.

Field contract of the reference's `DensityFeature` (mad/DensityFeature.py:5-84):
the detector fills `index, oct_scale (0 = upsampled, 1 = base), coords (integer
voxel in its own octave), map_coords, subv_map_coords (Angstrom), voxel_val`;
orientation adds `eqsp_size, box_size, box_side, main_bin, sec_bin, list_bins,
list_sec_bins, to_dom_mat, adj_sec_mat, Rfinal, ar_count`; description adds
`subeqsp_size, lin_ar_subeqsp`.  The 17^3 work arrays the reference keeps on each
record (grad_box, magn_box, weight_mask) live in LDS inside the HIP kernel and
are not materialised here.  The VMD/TCL debug printers are out of scope.
"""
import numpy as np


class DensityFeature(object):
    def __init__(self):
        # detector
        self.index = -1
        self.voxel_val = 0
        self.oct_scale = -1
        self.coords = []
        self.map_coords = []
        self.subv_map_coords = []
        self.ratio = 0
        # orientation
        self.eqsp_size = -1
        self.ar_count = []
        self.main_bin = -1
        self.sec_bin = -1
        self.list_bins = []
        self.list_sec_bins = []
        self.to_dom_mat = []
        self.adj_sec_mat = []
        self.Rfinal = []
        # descriptor
        self.subeqsp_size = -1
        self.lin_ar_subeqsp = []

    def set_detector_info(self, index, oct_scale, coords, map_coords, subv_map_coords, voxel_val):
        self.index = index
        self.oct_scale = oct_scale
        self.coords = coords
        self.map_coords = map_coords
        self.subv_map_coords = subv_map_coords
        self.voxel_val = voxel_val

    def set_orientator_info(self, eqsp_size, radius):
        self.eqsp_size = eqsp_size
        self.box_size = radius * 2 + 1
        self.box_side = radius
        self.ar_count = np.zeros(eqsp_size, dtype=np.int32)

    def set_descriptor_info(self, subeqsp_size, radius):
        self.subeqsp_size = subeqsp_size
        self.box_size = radius * 2 + 1
        self.box_side = radius

    def set_from_file_ori(self, index, main_bin, sec_bin, oct_scale, eqsp_size,
                          coord, map_coord, subv_map_coord, Rfinal, ar_count):
        self.index, self.main_bin, self.sec_bin = index, main_bin, sec_bin
        self.oct_scale, self.eqsp_size = oct_scale, eqsp_size
        self.coords, self.map_coords, self.subv_map_coords = coord, map_coord, subv_map_coord
        self.Rfinal = Rfinal
        self.ar_count = ar_count

    def set_from_file_dsc(self, index, main_bin, sec_bin, oct_scale, eqsp_size, subeqsp_size,
                          coord, map_coord, subv_map_coord, Rfinal, descr):
        self.set_from_file_ori(index, main_bin, sec_bin, oct_scale, eqsp_size,
                               coord, map_coord, subv_map_coord, Rfinal, self.ar_count)
        self.subeqsp_size = subeqsp_size
        self.lin_ar_subeqsp = descr

    def clone(self):
        """Shallow per-row copy (the reference deep-copies ~240 KB per row, Orientator.py:91,101)."""
        c = DensityFeature()
        c.__dict__.update(self.__dict__)
        return c

    def show(self):
        print("DF @o=%i: idx=%i main_bin=%i sec_bin=%i (base %i)" % (self.oct_scale, self.index, self.main_bin, self.sec_bin, self.eqsp_size))
        for name in ("coords", "map_coords", "subv_map_coords"):
            v = getattr(self, name)
            print("> %-16s %.3f %.3f %.3f" % (name, v[0], v[1], v[2]))
