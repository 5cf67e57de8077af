"""ctypes front-end of the CPU oracle (oracle/mad_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package (mad_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """-> path of the oracle library.  MAD_ORACLE_SANITIZE=1 selects the AddressSanitizer + UBSan build (`make asan`): the
    interpreter must then run with libasan preloaded (tests/test_oracle_sanitized.py does that in a child process)."""
    asan = os.environ.get("MAD_ORACLE_SANITIZE", "0") == "1"
    so = os.path.join(_HERE, "libmad_oracle_asan.so" if asan else "libmad_oracle.so")
    src = os.path.join(_HERE, "mad_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + (["asan"] if asan else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_ccc.restype = C.c_double
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _opt(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def belt_first(bounds):
    bounds = _c(bounds, np.float64)
    out = np.zeros(len(bounds), np.int32)
    lib().orc_eqsp_belt_first(_opt(bounds), C.c_int(len(bounds)), _opt(out))
    return out


def to_dom_mat(centers, a):
    centers = _c(centers, np.float64)
    m = np.zeros(9)
    lib().orc_to_dom_mat(_opt(centers), C.c_int(a), _opt(m))
    return m.reshape(3, 3)


def adj_sec_mat(bounds, centers, s):
    bounds = _c(bounds, np.float64)
    centers = _c(centers, np.float64)
    m = np.zeros(9)
    lib().orc_adj_sec_mat(_opt(bounds), _opt(centers), C.c_int(len(bounds)), C.c_int(s), _opt(m))
    return m.reshape(3, 3)


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.sqrt(np.dot(v, v))                                   # math_utils.py:5-13


def _rod(axis, angle):
    """math_utils.py:15-27, element for element."""
    axis = np.asarray(axis, dtype=np.float64)
    a = np.cos(angle / 2.0)
    b, c, d = -axis * np.sin(angle / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])


_tables = {}


def matrix_tables(bounds, centers):
    """to_dom_mat / adj_sec_mat of every zone with the reference's numpy expressions (eqsp.py:31-33 for the cartesian centres,
    Orientator.py:198-205 and :253-263): the C functions then use these bits instead of their own scalar restatement."""
    key = (bounds.tobytes(), centers.tobytes())
    if key not in _tables:
        Z = len(bounds)
        from math import sin, cos
        dom, adj = np.zeros((Z, 3, 3)), np.zeros((Z, 3, 3))
        first = belt_first(bounds)
        for a in range(Z):
            pol = centers[a]
            cart = np.array([sin(pol[1]) * cos(pol[0]), sin(pol[1]) * sin(pol[0]), cos(pol[1])])      # eqsp.py:31-33
            if a == 0:
                dom[a] = np.identity(3)                                                               # Orientator.py:211
            else:
                c = _unit(cart)
                angle = np.arccos(np.clip(np.dot(c, [0, 0, 1]), -1.0, 1.0))
                dom[a] = _rod(_unit(np.cross(c, [0, 0, 1])), angle)
            adj[a] = _rod([0, 0, 1], -1 * (centers[a][0] - centers[first[a]][0]))
        _tables[key] = (np.ascontiguousarray(dom.reshape(Z, 9)), np.ascontiguousarray(adj.reshape(Z, 9)))
    return _tables[key]


def orient(gx, gy, gz, octave, coords, bounds, centers, r=8, lim_main=6, lim_sec=6, want_counts=True, gw_sig=0.0):
    gx, gy, gz = _c(gx, np.float32), _c(gy, np.float32), _c(gz, np.float32)
    coords = _c(coords, np.int32).reshape(-1, 3)
    bounds, centers = _c(bounds, np.float64), _c(centers, np.float64)
    Z = len(bounds)
    n = len(coords)
    cap = max(1, n * lim_main * lim_sec)
    ra, rm, rs = (np.zeros(cap, np.int32) for _ in range(3))
    R = np.zeros((cap, 9))
    cnt = np.zeros((cap, Z), np.int32) if want_counts else None
    nrows = C.c_int64(0)
    nrej = C.c_int32(0)
    nx, ny, nz = gx.shape
    dom, adj = matrix_tables(bounds, centers)
    lib().orc_set_matrix_tables(_opt(dom), _opt(adj), C.c_int(Z))
    args = (_opt(gx), _opt(gy), _opt(gz), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(octave),
            _opt(coords), C.c_int(n), C.c_int(r), C.c_int(lim_main), C.c_int(lim_sec),
            _opt(bounds), _opt(centers), C.c_int(Z),
            _opt(ra), _opt(rm), _opt(rs), _opt(R), _opt(cnt), C.byref(nrows), C.c_int64(cap), C.byref(nrej))
    rc = lib().orc_orient_gw(*args, C.c_double(gw_sig)) if gw_sig else lib().orc_orient(*args)
    assert rc == 0
    k = nrows.value
    return dict(anchor=ra[:k].copy(), main=rm[:k].copy(), sec=rs[:k].copy(), R=R[:k].reshape(k, 3, 3).copy(),
                counts=None if cnt is None else cnt[:k].copy(), n_reject=nrej.value)


def describe(gx, gy, gz, octave, coords, R, bounds, r=8, dsc_size=64):
    gx, gy, gz = _c(gx, np.float32), _c(gy, np.float32), _c(gz, np.float32)
    coords = _c(coords, np.int32).reshape(-1, 3)
    R = _c(R, np.float64).reshape(-1, 9)
    bounds = _c(bounds, np.float64)
    Z = len(bounds)
    n = len(coords)
    out = np.zeros((n, dsc_size * Z), np.int16)
    nx, ny, nz = gx.shape
    rc = lib().orc_describe_sized(_opt(gx), _opt(gy), _opt(gz), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(octave),
                                  _opt(coords), _opt(R), C.c_int64(n), C.c_int(r), _opt(bounds), C.c_int(Z), C.c_int(dsc_size), _opt(out))
    assert rc == 0
    return out


def correlate(hi, lo, cc, want_scores=False):
    hi, lo = _c(hi, np.int16), _c(lo, np.int16)
    n_hi, D = hi.shape
    n_lo = lo.shape[0]
    scores = np.zeros((n_hi, n_lo)) if want_scores else None
    cap = max(1, n_hi * n_lo)
    ph, pl, ps = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    npairs = C.c_int64(0)
    rc = lib().orc_correlate(_opt(hi), C.c_int64(n_hi), _opt(lo), C.c_int64(n_lo), C.c_int(D), C.c_double(cc),
                             _opt(scores), _opt(ph), _opt(pl), _opt(ps), C.byref(npairs), C.c_int64(cap))
    assert rc == 0
    k = npairs.value
    return ph[:k].copy(), pl[:k].copy(), ps[:k].copy(), scores


def pose_score(pair_hi, pair_lo, pair_score, hi_p, hi_R, hi_meta, lo_p, lo_R, lo_meta, hi_cloud, lo_cloud, dist=4.0):
    pair_hi, pair_lo = _c(pair_hi, np.int32), _c(pair_lo, np.int32)
    pair_score = _c(pair_score, np.float64)
    hi_p, lo_p = _c(hi_p, np.float64), _c(lo_p, np.float64)
    hi_R, lo_R = _c(hi_R, np.float64).reshape(-1, 9), _c(lo_R, np.float64).reshape(-1, 9)
    hi_meta, lo_meta = _c(hi_meta, np.int32), _c(lo_meta, np.int32)
    hi_cloud, lo_cloud = _c(hi_cloud, np.float64), _c(lo_cloud, np.float64)
    n = len(pair_hi)
    res = np.zeros((n, 23))
    cnt = np.zeros(n, np.int32)
    rc = lib().orc_pose_score(_opt(pair_hi), _opt(pair_lo), _opt(pair_score), C.c_int64(n),
                              _opt(hi_p), _opt(hi_R), _opt(hi_meta), _opt(lo_p), _opt(lo_R), _opt(lo_meta),
                              _opt(hi_cloud), C.c_int64(len(hi_cloud)), _opt(lo_cloud), C.c_int64(len(lo_cloud)),
                              C.c_double(dist), _opt(res), _opt(cnt))
    assert rc == 0
    return res, cnt


def topk(counts, k):
    counts = _c(counts, np.int32)
    k = min(k, len(counts))
    order = np.zeros(k, np.int64)
    lib().orc_topk(_opt(counts), C.c_int64(len(counts)), C.c_int64(k), _opt(order))
    return order


def refine(grid, origin, vs, coords, n_steps=500, max_step=0.5, min_step=0.01, want_trace=False):
    grid = _c(grid, np.float32)
    coords = _c(coords, np.float64).copy()
    nx, ny, nz = grid.shape
    conv, last = C.c_int32(0), C.c_int32(0)
    trace = np.full((n_steps, 13), np.nan) if want_trace else None
    lib().orc_refine(_opt(grid), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                     C.c_double(origin[0]), C.c_double(origin[1]), C.c_double(origin[2]), C.c_double(vs),
                     _opt(coords), C.c_int64(len(coords)), C.c_int(n_steps), C.c_double(max_step), C.c_double(min_step),
                     C.byref(conv), C.byref(last), _opt(trace))
    return coords, bool(conv.value), last.value, trace


def splat(atoms, mass, vs, pad=0):
    atoms, mass = _c(atoms, np.float64), _c(mass, np.float64)
    dims = np.zeros(3, np.int32)
    mn = np.zeros(3)
    lib().orc_splat(_opt(atoms), _opt(mass), C.c_int64(len(atoms)), C.c_double(vs), C.c_int(pad), _opt(dims), _opt(mn), None)
    grid = np.zeros(int(dims[0]) * int(dims[1]) * int(dims[2]))
    lib().orc_splat(_opt(atoms), _opt(mass), C.c_int64(len(atoms)), C.c_double(vs), C.c_int(pad), _opt(dims), _opt(mn), _opt(grid))
    return grid, dims, mn


def blur(splat_grid, dims, resolution, vs, isovalue=0.0):
    splat_grid = _c(splat_grid, np.float64)
    dims = _c(dims, np.int32)
    r = C.c_int32(0)
    lib().orc_blur(_opt(splat_grid), _opt(dims), C.c_double(resolution), C.c_double(vs), C.c_double(isovalue), None, C.byref(r))
    shape = tuple(int(d) + 2 * r.value for d in dims)
    out = np.zeros(shape, np.float32)
    lib().orc_blur(_opt(splat_grid), _opt(dims), C.c_double(resolution), C.c_double(vs), C.c_double(isovalue), _opt(out), C.byref(r))
    return out, r.value


def structure_to_density(atoms, mass, resolution, vs, isovalue=0.0, pad=0):
    """PDB.structure_to_density (PDB.py:131-208): (grid f32[x][y][z], x0, y0, z0)."""
    g, dims, mn = splat(atoms, mass, vs, pad)
    out, r = blur(g, dims, resolution, vs, isovalue)
    margin = 2 + pad
    o = [mn[d] - (r + margin) * vs for d in range(3)]
    return out, o[0], o[1], o[2]


def ccc(g1, o1, g2, o2, vs, isovalue=0.0):
    """Dmap.get_CCC_with_grid; g1 and g2 are clamped in place like the reference."""
    assert g1.dtype == np.float32 and g2.dtype == np.float32 and g1.flags.c_contiguous and g2.flags.c_contiguous
    d1, d2 = np.array(g1.shape, np.int32), np.array(g2.shape, np.int32)
    o1, o2 = _c(o1, np.float64), _c(o2, np.float64)
    return lib().orc_ccc(_opt(g1), _opt(d1), _opt(o1), _opt(g2), _opt(d2), _opt(o2), C.c_double(vs), C.c_double(isovalue))


def _common_box(n1, a, n2, b):
    """One axis of the box shared by two grids that start at a and b (voxel units) and are n1 / n2 long:
    (start in grid 1, stop in grid 1, start in grid 2, stop in grid 2), structure_utils.py:181-238."""
    lo1, lo2 = (0, int(round(a - b))) if a > b else ((int(round(b - a)), 0) if a < b else (0, 0))
    if a + n1 > b + n2:
        hi1, hi2 = int(round(b + n2 - a)), int(round(n2))
    elif a + n1 < b + n2:
        hi1, hi2 = int(round(n1)), int(round(a + n1 - b))
    else:
        hi1, hi2 = int(round(n1)), int(round(n2))
    return lo1, hi1, lo2, hi2


def overlap(g1, o1, g2, o2, vs, isovalue=1e-8):
    """structure_utils.get_overlap (structure_utils.py:163-259): both grids are clamped in place; returns
    (#voxels of the common box occupied in both, #occupied voxels of g1) -- the reference's value is their ratio."""
    g1[g1 < isovalue] = 0
    g2[g2 < isovalue] = 0
    sl1, sl2 = [], []
    for d in range(3):
        lo1, hi1, lo2, hi2 = _common_box(g1.shape[d], o1[d] / vs, g2.shape[d], o2[d] / vs)
        if hi1 - lo1 < 0:      # :241-243
            return 0, int(np.count_nonzero(g1 > 0))
        sl1.append(slice(lo1, hi1))
        sl2.append(slice(lo2, hi2))
    a, b = g1[tuple(sl1)], g2[tuple(sl2)]
    ext = tuple(slice(0, min(x, y)) for x, y in zip(a.shape, b.shape))
    return int(np.count_nonzero((a[ext] > 0) & (b[ext] > 0))), int(np.count_nonzero(g1 > 0))


def overlap_ratio(g1, o1, g2, o2, vs, isovalue=1e-8):
    common, occupied = overlap(g1, o1, g2, o2, vs, isovalue)
    return common / occupied if occupied else 0


# ---------------------------------------------------------------------------------------------
# The same four stages on several host cores: the independent units (anchors, rows, hi rows, pairs) are
# cut into contiguous chunks, every chunk goes through the scalar C function above in its own thread
# (ctypes releases the GIL for the call) and the pieces are joined in order -- results are identical to the
# single-threaded functions.  Used by bench.py's all-cores CPU baseline only.
# ---------------------------------------------------------------------------------------------

def _chunks(n, parts):
    parts = max(1, min(parts, n))
    edges = [n * i // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _pool_map(fn, items, threads):
    if threads <= 1 or len(items) <= 1:
        return [fn(it) for it in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(fn, items))


def orient_mt(gx, gy, gz, octave, coords, bounds, centers, threads, **kw):
    coords = _c(coords, np.int32).reshape(-1, 3)
    parts = _pool_map(lambda ab: orient(gx, gy, gz, octave, coords[ab[0]:ab[1]], bounds, centers, **kw), _chunks(len(coords), 4 * threads), threads)
    spans = _chunks(len(coords), 4 * threads)
    out = dict(anchor=np.concatenate([p["anchor"] + a for p, (a, _) in zip(parts, spans)]) if parts else np.zeros(0, np.int32),
               n_reject=sum(p["n_reject"] for p in parts))
    for key in ("main", "sec", "R"):
        out[key] = np.concatenate([p[key] for p in parts]) if parts else np.zeros(0)
    out["counts"] = None if not parts or parts[0]["counts"] is None else np.concatenate([p["counts"] for p in parts])
    return out


def describe_mt(gx, gy, gz, octave, coords, R, bounds, threads, **kw):
    coords = _c(coords, np.int32).reshape(-1, 3)
    R = _c(R, np.float64).reshape(-1, 3, 3)
    parts = _pool_map(lambda ab: describe(gx, gy, gz, octave, coords[ab[0]:ab[1]], R[ab[0]:ab[1]], bounds, **kw), _chunks(len(coords), 4 * threads), threads)
    return np.concatenate(parts) if parts else np.zeros((0, 64 * len(bounds)), np.int16)


def correlate_mt(hi, lo, cc, threads):
    hi, lo = _c(hi, np.int16), _c(lo, np.int16)
    spans = _chunks(len(hi), 4 * threads)
    parts = _pool_map(lambda ab: correlate(hi[ab[0]:ab[1]], lo, cc), spans, threads)
    if not parts:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), None
    return (np.concatenate([p[0] + a for p, (a, _) in zip(parts, spans)]).astype(np.int32), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), None)


def pose_score_mt(pair_hi, pair_lo, pair_score, hi_p, hi_R, hi_meta, lo_p, lo_R, lo_meta, hi_cloud, lo_cloud, dist, threads):
    spans = _chunks(len(pair_hi), 4 * threads)
    parts = _pool_map(lambda ab: pose_score(pair_hi[ab[0]:ab[1]], pair_lo[ab[0]:ab[1]], pair_score[ab[0]:ab[1]], hi_p, hi_R, hi_meta,
                                            lo_p, lo_R, lo_meta, hi_cloud, lo_cloud, dist), spans, threads)
    if not parts:
        return np.zeros((0, 23)), np.zeros(0, np.int32)
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
