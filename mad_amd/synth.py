"""Deterministic synthetic inputs (tests, smoke, bench): pseudo-atom globules, small
density volumes and gradient fields.  numpy only; nothing here is on the hot path.

The generator follows SURVEY.md section 8(d): a subunit is a reflected random walk of
pseudo-atoms (step 1.5 A inside a sphere of radius R) with atom names cycling
N / CA / C / O; an assembly places rigid copies on a jittered lattice.
"""
import numpy as np

ATOM_CYCLE = (("N", "N"), ("CA", "C"), ("C", "C"), ("O", "O"))      # (name, element)
MASS = {"H": 1.00797, "C": 12.011, "N": 14.0067, "O": 15.9994, "S": 32.06, "P": 30.97376}


def random_globule(n_atoms, radius, seed, step=1.5):
    """Reflected random walk: (coords float64 [n,3] rounded to the PDB's 3 decimals, names, elements)."""
    rng = np.random.default_rng(seed)
    pts = np.zeros((n_atoms, 3))
    p = np.zeros(3)
    for i in range(n_atoms):
        d = rng.normal(size=3)
        d *= step / np.linalg.norm(d)
        q = p + d
        if np.linalg.norm(q) > radius:
            q = p - d
            if np.linalg.norm(q) > radius:
                q = p * (radius - step) / max(np.linalg.norm(p), 1e-9)
        p = q
        pts[i] = p
    names = [ATOM_CYCLE[i % 4][0] for i in range(n_atoms)]
    elems = [ATOM_CYCLE[i % 4][1] for i in range(n_atoms)]
    return np.round(pts, 3), names, elems


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def place(coords, R, t):
    """Rigid placement in the reference's row-vector convention: (x - centroid) @ R + t."""
    return np.round((coords - coords.mean(axis=0)) @ R + np.asarray(t), 3)


def masses(elems):
    return np.array([MASS.get(e.upper(), MASS["C"]) for e in elems])


def write_pdb(path, coords, names, elems, chain="A"):
    """Minimal PDB v3.3 ATOM records (the columns the reference's reader uses, PDB.py:41-65)."""
    with open(path, "w") as f:
        for i, (c, nm, el) in enumerate(zip(coords, names, elems)):
            name = nm if len(nm) == 4 else " %-3s" % nm
            f.write("%-6s%5i %4s %3s %1s%4i    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n"
                    % ("ATOM", (i + 1) % 100000, name, "GLY", chain, (i // 4 + 1) % 10000, c[0], c[1], c[2], 1.0, 0.0, el))


def write_situs(path, grid, origin, voxsp):
    """Situs text volume as the reference reads it (Dmap.py:13-24): x fastest."""
    g = np.asarray(grid)
    with open(path, "w") as f:
        f.write("%f %f %f %f %i %i %i\n\n" % (voxsp, origin[0], origin[1], origin[2], g.shape[0], g.shape[1], g.shape[2]))
        flat = g.reshape(-1, order="F")
        for i in range(0, len(flat), 10):
            f.write("".join("   %10.6f   " % v for v in flat[i:i + 10]) + "\n")


def blob_volume(shape, n_blobs, seed, sigma=(2.0, 4.0), hollow=0.0):
    """Smooth positive float32 volume: a sum of isotropic Gaussians, optionally with a box of exact zeros."""
    rng = np.random.default_rng(seed)
    ax = [np.arange(s, dtype=np.float64) for s in shape]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    vol = np.zeros(shape)
    for _ in range(n_blobs):
        c = rng.uniform(0.15, 0.85, size=3) * np.array(shape)
        s = rng.uniform(*sigma)
        a = rng.uniform(0.3, 1.0)
        vol += a * np.exp(-((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) / (2 * s * s))
    vol /= vol.max()
    if hollow > 0:
        k = [int(s * hollow) for s in shape]
        vol[:k[0], :k[1], :] = 0.0
    return vol.astype(np.float32)


def gradient_field(vol):
    """np.gradient of a float32 volume as the reference stores it: a (X, Y, Z, 3) float32 view."""
    return np.moveaxis(np.array(np.gradient(vol.astype(np.float32))), 0, -1)


def interior_anchors(shape, n, margin, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(margin, s - margin, size=n) for s in shape], axis=1).astype(np.int32)
