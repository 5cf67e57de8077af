"""Atomic structure object of the hot path.

Same public surface as the reference's `PDB` (mad/PDB.py:7-292): `coords` (float64
[n,3]), `info` rows `[serial, name, resname, chain, resnum, element, record]`,
`CA_idx`, `BB_idx`, `n_atoms`, rigid transforms in the row-vector convention
(`coords @ R`, PDB.py:109-113), RMSDs and `structure_to_density`.  The density
simulation (trilinear mass splat + Gaussian blur + normalise, PDB.py:131-292) runs on
the GPU through `mad_structure_to_density`; everything else is bookkeeping.
"""
import os
import sys

import numpy as np

from . import _lib

# PDB.py:220-221
MASS = {"H": 1.00797, "BE": 9.01218, "C": 12.011, "N": 14.0067, "O": 15.9994, "F": 18.998403, "S": 32.06,
        "P": 30.97376, "MG": 24.305, "CL": 35.453, "K": 39.0983, "CA": 40.078, "MN": 54.9380, "FE": 55.847,
        "NI": 58.70, "CU": 63.546, "ZN": 65.38, "SE": 78.96}


class PDB(object):
    def __init__(self, pdb_file):
        self.pdb_file = pdb_file
        if not os.path.exists(pdb_file):
            print("PDB> File not found: %s" % pdb_file)
            sys.exit(1)
        coords, info, ca, bb = [], [], [], []
        # Column layout of PDB v3.3 coordinate records; a field that fails to parse keeps the
        # previous atom's value, as in the reference (PDB.py:45-57).
        serial = resnum = 0
        name = resname = chain = elem = ""
        x = y = z = 0.0
        with open(pdb_file, "r") as fh:
            for line in fh:
                rec = line[0:6].strip()
                if rec not in ("ATOM", "HETATM"):
                    continue
                try:
                    serial = int(line[6:11].strip())
                    name = line[12:16].strip()
                    resname = line[17:20]
                    chain = line[21]
                    resnum = int(line[22:26].strip())
                    x, y, z = float(line[30:38]), float(line[38:46]), float(line[46:54])
                    elem = line[76:78].strip()
                except Exception:
                    pass
                idx = len(coords)
                info.append([serial, name, resname, chain, resnum, elem, rec])
                coords.append([x, y, z])
                if name == "CA":
                    ca.append(idx)
                if name in ("C", "CA", "N", "O"):
                    bb.append(idx)
        self.info = info
        self.coords = np.array(coords)
        self.CA_idx = tuple(ca)
        self.BB_idx = bb
        self.n_atoms = len(self.coords)
        self.n_CA = len(self.CA_idx)
        self.minx, self.miny, self.minz = np.amin(self.coords, axis=0)
        self.maxx, self.maxy, self.maxz = np.amax(self.coords, axis=0)

    def write_pdb(self, outname):
        with open(outname, "w") as out:
            for i in range(self.n_atoms):
                serial, name, resname, chain, resnum, elem, rec = self.info[i]
                # 4-letter atom names start in column 13, shorter ones in column 14 (PDB.py:85-90)
                atom = "%-4s" % name if len(name) == 4 else " %-3s" % name
                cx, cy, cz = self.coords[i]
                out.write("%-6s%5i %s %3s%2s%4s    %8.3f%8.3f%8.3f%6.2f%6.2f          %-2s\n"
                          % (rec, serial, atom, resname, chain, resnum, cx, cy, cz, 1.0, 0.0, elem))

    def rgyr(self):
        d = self.coords - np.mean(self.coords, axis=0)
        return np.sqrt(np.sum(d ** 2) / self.coords.shape[0])

    # -- rigid-body manipulation ---------------------------------------------------------
    def get_coords(self):
        return self.coords

    def set_coords(self, coords):
        self.coords = coords.copy()

    def rotate_atoms(self, rot_mat):
        self.coords = np.dot(self.coords, rot_mat)

    def translate_atoms(self, trans_vec):
        self.coords += np.array(trans_vec)

    def get_rmsd_with(self, pdb):
        d = np.square(self.coords - pdb.coords)
        return np.sqrt(np.sum(d, axis=(0, 1)) / d.shape[0])

    def get_rmsdCA_with(self, pdb):
        if not len(self.CA_idx):
            print("PDB> No alpha carbons detected; returning all-atom RMSD instead.")
            return self.get_rmsd_with(pdb)
        d = np.square(self.coords[self.CA_idx, :] - pdb.coords[pdb.CA_idx, :])
        return np.sqrt(np.sum(d, axis=(0, 1)) / d.shape[0])

    # -- density ---------------------------------------------------------------------------
    def atom_masses(self):
        """Per-atom mass by element symbol, carbon when unknown (PDB.py:225-233)."""
        out = np.empty(self.n_atoms)
        for i, row in enumerate(self.info):
            el = row[-2].upper()
            if el not in MASS:
                print("PDB> (dens) Element %s not in dict. Using mass of carbon." % el)
                el = "C"
            out[i] = MASS[el]
        return out

    def structure_to_density(self, resolution, voxelsp, isovalue=0.0, pad=0, outname=""):
        """(grid float32 [x,y,z], x0, y0, z0) -- PDB.py:131-208, computed on the GPU."""
        grid, x0, y0, z0 = _lib.get_lib().structure_to_density(self.coords, self.atom_masses(), resolution, voxelsp,
                                                              isovalue=isovalue, pad=pad)
        if outname != "":
            from . import mapio
            mapio.write_volume(outname, grid, (x0, y0, z0), voxelsp)
        return grid, x0, y0, z0
