"""Volume file formats of the path's two ends: Situs text maps and MRC2014 (mode 2).

The reference reads maps through `mrcfile` and a hand-rolled Situs parser
(mad/Dmap.py:11-43, mad/MapSpace.py:77-113) and writes them in mad/Dmap.py:377-415 /
mad/PDB.py:165-206.  `mrcfile` is not a dependency here: the header fields the
reference touches (nx..nz, mode, n*start, mx..mz, cella, mapc/r/s, origin) are read
and written directly.  Out of the hot path.
"""
import struct

import numpy as np


def read_situs(path, dtype=np.float32):
    """-> (grid [x,y,z], voxsp, (xi, yi, zi)); Situs stores x fastest (Dmap.py:17-24)."""
    with open(path, "r") as fh:
        header = fh.readline().replace("\n", "").replace("  ", "").split(" ")
        fh.readline()
        voxsp, xi, yi, zi = [float(v) for v in header[:4]]
        xb, yb, zb = [int(v) for v in header[4:7]]
        data = np.array(fh.read().split(), dtype=np.float64)
    grid = np.reshape(data.astype(dtype), (xb, yb, zb), order="F")
    return grid, voxsp, (xi, yi, zi)


def write_situs(path, grid, origin, voxsp):
    g = np.asarray(grid)
    with open(path, "w") as fh:
        fh.write("%f %f %f %f %i %i %i\n\n" % (voxsp, origin[0], origin[1], origin[2], g.shape[0], g.shape[1], g.shape[2]))
        # Dmap.py:382-389, character for character: x fastest, "   %6.6f " per voxel, a line break after every tenth
        flat = g.reshape(-1, order="F")
        for i in range(0, len(flat), 10):
            chunk = flat[i:i + 10]
            fh.write("".join("   %6.6f " % v for v in chunk) + ("\n" if len(chunk) == 10 else ""))


_MRC_DTYPES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}


def read_mrc(path):
    """-> dict(data [z-ish,y-ish,x-ish as stored], mapc/r/s, voxel size, nstart, origin, mxyz)."""
    with open(path, "rb") as fh:
        head = fh.read(1024)
        nx, ny, nz, mode, nxs, nys, nzs, mx, my, mz = struct.unpack("<10i", head[:40])
        endian = "<"
        if not (0 < nx < 65536 and 0 < ny < 65536 and 0 < nz < 65536):      # big-endian file
            endian = ">"
            nx, ny, nz, mode, nxs, nys, nzs, mx, my, mz = struct.unpack(">10i", head[:40])
        cella = struct.unpack(endian + "3f", head[40:52])
        mapc, mapr, maps = struct.unpack(endian + "3i", head[64:76])
        nsymbt = struct.unpack(endian + "i", head[92:96])[0]
        origin = struct.unpack(endian + "3f", head[196:208])
        if mode not in _MRC_DTYPES:
            raise ValueError("MaD> unsupported MRC mode %d in %s" % (mode, path))
        fh.seek(1024 + max(nsymbt, 0))
        data = np.frombuffer(fh.read(), dtype=np.dtype(_MRC_DTYPES[mode]).newbyteorder(endian), count=nx * ny * nz)
    data = data.reshape(nz, ny, nx).astype(np.float32)
    vox = (cella[0] / mx if mx else 1.0, cella[1] / my if my else 1.0, cella[2] / mz if mz else 1.0)
    return dict(data=data, mapc=mapc, mapr=mapr, maps=maps, voxel_size=vox, nstart=(nxs, nys, nzs), origin=origin,
                mxyz=(mx, my, mz))


def load_mrc_as_xyz(path):
    """The reference's reading convention (Dmap.py:27-43): -> (grid [x,y,z] float32, voxsp, (xi, yi, zi), (xb, yb, zb))."""
    m = read_mrc(path)
    axis_order = [m["mapc"] - 1, m["mapr"] - 1, m["maps"] - 1]
    voxsp = m["voxel_size"][0]
    if np.all(m["nstart"]):
        start = np.array(m["nstart"], dtype=int)
        org = [start[a] * voxsp for a in axis_order]
    else:
        start = np.array(m["origin"], dtype=int)      # truncation to int is the reference's (Dmap.py:38)
        org = [start[a] for a in axis_order]
    box = np.array(m["mxyz"], dtype=int)
    dims = [box[a] for a in axis_order]
    grid = np.transpose(m["data"].copy(), axis_order[::-1])
    return grid, voxsp, tuple(float(v) for v in org), tuple(int(v) for v in dims)


def write_mrc(path, grid_xyz, origin, voxsp):
    """Mode-2 MRC2014 with mapc/r/s = 1/2/3 and the origin field set (Dmap.py:392-415, PDB.py:183-206)."""
    g = np.ascontiguousarray(np.asarray(grid_xyz, dtype=np.float32).transpose(2, 1, 0))
    nz, ny, nx = g.shape
    head = bytearray(1024)
    struct.pack_into("<10i", head, 0, nx, ny, nz, 2, 0, 0, 0, nx, ny, nz)
    struct.pack_into("<6f", head, 40, nx * voxsp, ny * voxsp, nz * voxsp, 90.0, 90.0, 90.0)
    struct.pack_into("<3i", head, 64, 1, 2, 3)
    struct.pack_into("<3f", head, 76, float(g.min()), float(g.max()), float(g.mean()))
    struct.pack_into("<3f", head, 196, origin[0], origin[1], origin[2])
    head[208:212] = b"MAP "
    head[212:216] = bytes([0x44, 0x44, 0, 0])
    struct.pack_into("<f", head, 216, float(g.std()))
    with open(path, "wb") as fh:
        fh.write(bytes(head))
        fh.write(g.astype("<f4").tobytes())


def write_volume(path, grid_xyz, origin, voxsp):
    if path.lower().endswith((".sit", ".situs")):
        write_situs(path, grid_xyz, origin, voxsp)
    else:
        write_mrc(path, grid_xyz, origin, voxsp)
