"""Orientation assignment (hot path rows a1-a8).

Same constructor and entry point as the reference's `Orientator`
(mad/Orientator.py:13, :68): `Orientator(eqsp_size=112, main_ori=6, sec_ori=6,
ori_radius=16, gw_sig=0, magn_weighted=False).assign_orientations(ms, df_list)` returns
one DensityFeature per (anchor, main bin, secondary bin), in the reference's order,
with `main_bin, sec_bin, list_bins, list_sec_bins, to_dom_mat, adj_sec_mat, Rfinal,
ar_count` filled in.  The arithmetic (steps 01-05, Orientator.py:116-343) runs in the
HIP kernel `k_orient` through `mad_orient`; nothing is computed here.

Differences, all deliberate:
  * rows are shallow clones of their anchor (the reference deep-copies ~240 KB of work
    arrays per row, Orientator.py:91,101); the 17^3 work arrays never leave the GPU;
  * `step1_reject` is initialised (the reference's border-reject path raises
    AttributeError, Orientator.py:133);
  * `gw_sig` (Gaussian window, Orientator.py:49-54) runs on the device with zone sums in 2^-50 fixed point; `magn_weighted` is
    stored and, as in the reference (Orientator.py:33 is its only use), never read; `eqsp_size` 112 (default) and 16 (the
    reference's other table) -- goldens g17 / g18 from the reference.
"""
import numpy as np

from . import _lib
from .eqsp.eqsp import EQSP_Sphere
from .orient_tables import orientation_matrices


class Orientator(object):
    def __init__(self, eqsp_size=112, main_ori=6, sec_ori=6, ori_radius=16, gw_sig=0, magn_weighted=False):
        self.eqsp_size = eqsp_size
        self.eqsp = EQSP_Sphere(size=eqsp_size)
        self.lim_main_ori = main_ori
        self.lim_sec_ori = sec_ori
        self.cutoff_magn = 1e-5
        if ori_radius % 2:
            print("MaD> ERROR: radius %i invalid, must be even. Setting %i instead." % (ori_radius, ori_radius - 1))
            ori_radius -= 1
        self.ori_radius = ori_radius // 2
        self.gw_sig = gw_sig
        # the reference stores this flag and never reads it again (Orientator.py:33 is its only use): accepted, without effect
        self.magn_weighted = magn_weighted
        # sphere mask of Orientator.py:38-47, kept for inspection (the device builds the same one)
        dr = self.ori_radius
        g = np.mgrid[-dr:dr + 1, -dr:dr + 1, -dr:dr + 1]
        self.sphere_mask_ori_dict = (np.sqrt(np.sum(g * g, 0)) <= dr * 1.05).astype(int)
        # Orientator.py:49-54: Gaussian window times the mask (ones times the mask when gw_sig is 0), kept for inspection
        self.gauss_weight_ori_dict = (np.exp(-1 * np.divide(np.sum(g * g, 0), 2 * (gw_sig) ** 2)) if gw_sig else np.ones_like(np.sum(g * g, 0))) * self.sphere_mask_ori_dict
        self.to_dom_table, self.adj_sec_table = orientation_matrices(self.eqsp)
        self.step1_reject = 0
        self.time1 = self.time2 = self.time3 = self.time4 = self.time5 = 0

    def _bind(self, lib):
        key = ("ori", self.eqsp_size)
        if lib._eq_loaded.get(0) != key:
            lib.set_eqsp(0, self.eqsp.sphere_eqsp, self.to_dom_table, self.adj_sec_table)
            lib._eq_loaded[0] = key
        lib.set_orient_window(self.gw_sig)

    def assign_orientations(self, ms, df_list):
        print("MaD> Orienting %i anchors..." % (len(df_list)))
        lib = _lib.get_lib()
        self._bind(lib)
        try:
            return self._assign(lib, ms, df_list)
        finally:
            lib.set_orient_window(0.0)      # the window is state of the context: do not leave this object's behind

    def _assign(self, lib, ms, df_list):
        slots = ms.device_slots(lib)
        oriented = {}
        for octave in sorted(set(df.oct_scale for df in df_list)):
            ids = [i for i, df in enumerate(df_list) if df.oct_scale == octave]
            coords = np.array([df_list[i].coords for i in ids], dtype=np.int32).reshape(-1, 3)
            # list position == octave when both octaves exist; a single-octave space holds it at 0
            slot = slots[octave] if len(ms.grad_list) > 1 else slots[0]
            stride_octave = octave if len(ms.grad_list) > 1 else (1 if ms.oct_mode == "base" else 0)
            rows = lib.orient(slot, stride_octave, coords, r=self.ori_radius, lim_main=self.lim_main_ori,
                              lim_sec=self.lim_sec_ori, want_counts=True, Z=self.eqsp_size)
            self.step1_reject += rows["n_reject"]
            for k in range(len(rows["anchor"])):
                oriented.setdefault(ids[rows["anchor"][k]], []).append(k)
            oriented[("rows", octave)] = rows
            oriented[("ids", octave)] = ids
        out = []
        for i, df in enumerate(df_list):
            ks = oriented.get(i)
            if not ks:
                continue
            rows = oriented[("rows", df.oct_scale)]
            df.set_orientator_info(self.eqsp_size, self.ori_radius)
            mains = sorted(set(int(rows["main"][k]) for k in ks))
            for k in ks:
                row = df.clone()
                row.main_bin = int(rows["main"][k])
                row.sec_bin = int(rows["sec"][k])
                row.list_bins = np.array(mains)
                row.list_sec_bins = np.array([int(rows["sec"][j]) for j in ks if rows["main"][j] == rows["main"][k]])
                row.to_dom_mat = self.to_dom_table[row.main_bin].copy()
                row.adj_sec_mat = self.adj_sec_table[row.sec_bin].copy()
                row.Rfinal = rows["R"][k].copy()
                row.ar_count = rows["counts"][k].copy()
                out.append(row)
        return out

    def show_timing(self):
        print("MaD> Step timing: orientation runs on the GPU; use Lib.timing_get('orient')")
