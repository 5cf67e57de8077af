"""ctypes binding of libmad_amd.so (the C-ABI in include/mad_amd.h).

This is the ONLY compute back-end of the package: there is no CPU fallback.  If the
shared library is missing, or no gfx950 device can be opened, the first call raises
`MadBackendError` -- loudly, as a drop-in for a hot path must.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAD_LIB_PATH: another build of the same library (diagnostic builds with other compiler flags; tools/build_variant.sh)
LIB_PATH = os.environ.get("MAD_LIB_PATH") or os.path.join(_HERE, "libmad_amd.so")
RESULT_COLS = 23

ERRORS = {-22: "EINVAL", -12: "ENOMEM", -28: "ENOSPC", -19: "ENODEV", -33: "EDOM", -5: "EHIP"}

# every symbol include/mad_amd.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mad_init", "mad_destroy", "mad_last_error", "mad_synchronize", "mad_stream", "mad_set_overlap",
    "mad_timing_enable", "mad_timing_reset", "mad_timing_get", "mad_last_ms", "mad_probe_peaks",
    "mad_set_eqsp", "mad_upload_field", "mad_upload_field_device", "mad_free_field",
    "mad_set_orient_window", "mad_orient", "mad_describe", "mad_describe_sized", "mad_correlate", "mad_pose_score", "mad_topk",
    "mad_set_create", "mad_set_destroy", "mad_set_build", "mad_set_build_many", "mad_set_load", "mad_set_size", "mad_set_download",
    "mad_match_topk", "mad_match_topk_many", "mad_match_topk_many_begin", "mad_match_topk_many_finish", "mad_set_batching", "mad_set_option", "mad_last_pose_kernel", "mad_last_pose_selected", "mad_device_allocations", "mad_match_fetch", "mad_match_results", "mad_match_used",
    "mad_match_shard_pairs", "mad_match_shard_topk", "mad_match_shard_begin", "mad_match_shard_score", "mad_match_shard_record_doubles", "mad_match_shard_collect", "mad_match_shard_wait",
    "mad_set_wire_bytes", "mad_set_export", "mad_set_import", "mad_set_lane", "mad_set_stream", "mad_set_bind_lane",
    "mad_upload_density", "mad_refine", "mad_structure_to_density", "mad_ccc", "mad_density_ccc", "mad_dock_refine_score", "mad_grid_overlap", "mad_overlap_matrix",
    "mad_space_create", "mad_space_destroy", "mad_space_build", "mad_space_info", "mad_space_download",
    "mad_space_peaks", "mad_space_patches",
]


class MadBackendError(RuntimeError):
    pass


_dll = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 next to its extension modules.  Two HIP runtimes in one process do not share a
    device: whichever is loaded second finds "no HIP GPUs".  So when torch is installed, its copy of the runtime is loaded first
    and libmad_amd.so binds to that one (same soname), whatever the import order of torch and this package.  torch itself is
    neither imported nor needed."""
    import importlib.util
    if os.environ.get("MAD_OWN_HIP_RUNTIME", "0") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        path = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load_library():
    """dlopen the HIP library (works without a GPU; opening a device does not)."""
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise MadBackendError(
                "MaD> %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mad_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        _share_torch_hip_runtime()
        try:
            _dll = C.CDLL(LIB_PATH)
        except OSError as e:
            raise MadBackendError("MaD> cannot load %s: %s" % (LIB_PATH, e))
        _dll.mad_last_error.restype = C.c_char_p
        _dll.mad_last_error.argtypes = [C.c_void_p]
        _dll.mad_stream.restype = C.c_void_p
        _dll.mad_stream.argtypes = [C.c_void_p]
        _dll.mad_last_ms.restype = C.c_double
        _dll.mad_last_ms.argtypes = [C.c_void_p, C.c_char_p]
        _dll.mad_set_stream.restype = C.c_void_p
        _dll.mad_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        _dll.mad_last_pose_selected.restype = C.c_int64
        _dll.mad_last_pose_selected.argtypes = [C.c_void_p]
        _dll.mad_device_allocations.restype = C.c_int64
        _dll.mad_device_allocations.argtypes = [C.c_void_p]
        _dll.mad_set_wire_bytes.restype = C.c_int64
        _dll.mad_set_wire_bytes.argtypes = [C.c_int, C.c_int64]
        _dll.mad_match_shard_record_doubles.restype = C.c_int64
        _dll.mad_match_shard_record_doubles.argtypes = [C.c_int64]
        _dll.mad_destroy.restype = None
        _dll.mad_set_destroy.restype = None
        _dll.mad_space_destroy.restype = None
    return _dll


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _pad_rows(dsc, to=128):
    """Descriptor rows zero-padded to a multiple of 128 counts: the correlation kernel consumes K in steps of 128 bytes, and zeros
    change neither a dot product nor a norm (row lengths 432 and 16 of Descriptor(dsc_size=27 | 1))."""
    d = dsc.shape[1] if dsc.ndim == 2 else 0
    if d == 0 or d % to == 0:
        return dsc
    out = np.zeros((dsc.shape[0], (d + to - 1) // to * to), dsc.dtype)
    out[:, :d] = dsc
    return out


class BuildBatch(object):
    """The argument tables of one `mad_set_build_many` call (several structures, one launch per stage), kept alive between steps."""

    def __init__(self, lib, jobs, r, lim_main, lim_sec, gw_sig=0.0):
        self.lib, self.r, self.lim_main, self.lim_sec, self.gw_sig = lib, int(r), int(lim_main), int(lim_sec), float(gw_sig)
        n = len(jobs)
        self.sets = [job[5] if len(job) > 5 and job[5] is not None else DeviceSet(lib) for job in jobs]
        self._keep = []
        coords, octave, subv, index, counts, slots = [], [], [], [], [], []
        for job in jobs:
            c, o = _c(job[1], np.int32).reshape(-1, 3), _c(job[2], np.int32)
            v, i = _c(job[3], np.float64).reshape(-1, 3), _c(job[4], np.int32)
            if not (len(c) == len(o) == len(v) == len(i)):
                raise ValueError("set_build_many: anchor arrays of different lengths")
            self._keep += [c, o, v, i]
            coords.append(c.ctypes.data); octave.append(o.ctypes.data); subv.append(v.ctypes.data); index.append(i.ctypes.data)
            counts.append(len(o))
            slots += [int(job[0][0]), int(job[0][1])]
        self.counts = counts
        self._h = (C.c_void_p * max(n, 1))(*[s.h.value for s in self.sets])
        self._slots = (C.c_int * max(2 * n, 1))(*slots)
        self._coords = (C.c_void_p * max(n, 1))(*coords)
        self._octave = (C.c_void_p * max(n, 1))(*octave)
        self._subv = (C.c_void_p * max(n, 1))(*subv)
        self._index = (C.c_void_p * max(n, 1))(*index)
        self._n = (C.c_int * max(n, 1))(*counts)

    def run(self):
        self.lib.set_orient_window(self.gw_sig)      # the window is state of the context: every build states its own, as set_build does
        for s, n in zip(self.sets, self.counts):
            s.n_anchors = n
        self.lib._chk(self.lib.dll.mad_set_build_many(self.lib.ctx, C.c_int(len(self.sets)), self._h, self._slots, self._coords, self._octave,
                                                      self._subv, self._index, self._n, C.c_int(self.r), C.c_int(self.lim_main),
                                                      C.c_int(self.lim_sec)))
        return self.sets


class DeviceSet(object):
    """Device-resident oriented-anchor rows of one structure (mad_set)."""

    def __init__(self, lib, lane=None):
        self.lib = lib
        self.h = C.c_void_p()
        lib._chk(lib.dll.mad_set_create(lib.ctx, C.byref(self.h)))
        self.n_anchors = 0
        if lane is not None:
            self.bind_lane(lane)

    def lane(self):
        return int(self.lib.dll.mad_set_lane(self.lib.ctx, self.h))

    def bind_lane(self, lane):
        """Put the set on lane `lane` (0..7): sets of one lane run in order on one stream.  Synchronises the context."""
        self.lib._chk(self.lib.dll.mad_set_bind_lane(self.lib.ctx, self.h, C.c_int(int(lane))))

    def stream(self):
        """The HIP stream (as an integer handle) the set's kernels are enqueued on."""
        return self.lib.dll.mad_set_stream(self.lib.ctx, self.h)

    def size(self):
        n = C.c_int64(0)
        a = C.c_int32(0)
        self.lib._chk(self.lib.dll.mad_set_size(self.lib.ctx, self.h, C.byref(n), C.byref(a)))
        return n.value, a.value

    def download(self, want_dsc=True, D=1024):
        n, _ = self.size()
        out = dict(anchor=np.zeros(n, np.int32), main=np.zeros(n, np.int32), sec=np.zeros(n, np.int32),
                   R=np.zeros((n, 3, 3)), dsc=np.zeros((n, D), np.int16) if want_dsc else None)
        self.lib._chk(self.lib.dll.mad_set_download(self.lib.ctx, self.h, _p(out["anchor"]), _p(out["main"]), _p(out["sec"]),
                                                    _p(out["R"]), _p(out["dsc"])))
        return out

    def close(self):
        if self.h and self.lib.ctx:
            self.lib.dll.mad_set_destroy(self.lib.ctx, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSpace(object):
    """Device-resident scale space of one structure (mad_space): the volumes of MapSpace.build_space."""

    GRID, LOG, GAUSS = 0, 1, 2

    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p()
        lib._chk(lib.dll.mad_space_create(lib.ctx, C.byref(self.h)))
        self.shapes, self.kinds, self.dtypes = [], [], []

    def build(self, grid, pad=9, oct_mode="both", sig_init=2, sig_presmooth=1, slot_up=-1, slot_base=-1):
        """grid: float32 or float64 (X, Y, Z).  Fills the field slots with the gradient texels."""
        from . import scale_tables as st
        g = np.asarray(grid)
        if g.ndim != 3 or g.dtype not in (np.float32, np.float64):
            raise ValueError("density grid must be a 3-D float32 or float64 array")
        g = np.ascontiguousarray(g)
        mode = {"base": 1, "up": 2, "both": 3}[oct_mode]
        R = st.kernel_radius(sig_init)
        g0 = np.ascontiguousarray(st.gaussian_kernel1d(sig_init, 0, R)[::-1])
        g2 = np.ascontiguousarray(st.gaussian_kernel1d(sig_init, 2, R)[::-1])
        pre_R = st.kernel_radius(sig_presmooth) if (sig_presmooth and mode & 2) else 0
        pre = np.ascontiguousarray(st.gaussian_kernel1d(sig_presmooth, 0, pre_R)[::-1]) if pre_R else None
        P3, I3 = C.c_void_p * 3, C.c_void_p * 3
        lu = ev_w = ev_i = None
        keep = []
        if mode & 2:
            tabs = [st.spline_tables(int(n) + 2 * pad) for n in g.shape]
            keep = tabs
            lu = P3(*[t[0].ctypes.data for t in tabs])
            ev_w = P3(*[t[1].ctypes.data for t in tabs])
            ev_i = I3(*[t[2].ctypes.data for t in tabs])
        self.lib._chk(self.lib.dll.mad_space_build(
            self.lib.ctx, self.h, _p(g), C.c_int(1 if g.dtype == np.float64 else 0), C.c_int(g.shape[0]), C.c_int(g.shape[1]),
            C.c_int(g.shape[2]), C.c_int(pad), C.c_int(mode), _p(g0), _p(g2), C.c_int(R), C.c_double(float(sig_init) ** 2),
            _p(pre), C.c_int(pre_R), lu, ev_w, ev_i, C.c_int(slot_up), C.c_int(slot_base)))
        del keep
        n = C.c_int(0)
        dims, kind, f64 = np.zeros(6, np.int32), np.zeros(2, np.int32), np.zeros(2, np.int32)
        self.lib._chk(self.lib.dll.mad_space_info(self.lib.ctx, self.h, C.byref(n), _p(dims), _p(kind), _p(f64)))
        self.shapes = [tuple(int(v) for v in dims[3 * o:3 * o + 3]) for o in range(n.value)]
        self.kinds = [int(kind[o]) for o in range(n.value)]
        self.dtypes = [np.float64 if f64[o] else np.float32 for o in range(n.value)]
        return self

    def download(self, entry, what):
        out = np.empty(self.shapes[entry], self.dtypes[entry])
        self.lib._chk(self.lib.dll.mad_space_download(self.lib.ctx, self.h, C.c_int(entry), C.c_int(what), _p(out)))
        return out

    def peaks(self, entry, threshold=5e-2, border=12):
        """-> (coords int (n, 3), values float64 (n,)) in skimage's order: descending value, row-major among equals."""
        cap = 1 << 16
        while True:
            idx, val = np.zeros(cap, np.int64), np.zeros(cap, np.float64)
            n = C.c_int64(0)
            rc = self.lib.dll.mad_space_peaks(self.lib.ctx, self.h, C.c_int(entry), C.c_double(threshold), C.c_int(border), _p(idx), _p(val),
                                              C.c_int64(cap), C.byref(n))
            if rc == -28 and n.value > cap:      # MAD_ENOSPC: n holds the required capacity
                cap = int(n.value) + 1024
                continue
            self.lib._chk(rc)
            break
        idx, val = idx[:n.value], val[:n.value]
        order = np.argsort(idx, kind="stable")              # row-major, like np.nonzero
        idx, val = idx[order], val[order]
        order = np.argsort(-val, kind="stable")
        idx, val = idx[order], val[order]
        _, ny, nz = self.shapes[entry]
        coords = np.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], 1).astype(np.int64)
        return coords, val

    def patches(self, entry, coords, r):
        coords = _c(coords, np.int32).reshape(-1, 3)
        side = 2 * r + 1
        out = np.zeros((len(coords), side, side, side), self.dtypes[entry])
        self.lib._chk(self.lib.dll.mad_space_patches(self.lib.ctx, self.h, C.c_int(entry), _p(coords), C.c_int(len(coords)), C.c_int(r), _p(out)))
        return out

    def close(self):
        if self.h and self.lib.ctx:
            self.lib.dll.mad_space_destroy(self.lib.ctx, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lib(object):
    """One mad_ctx on one GPU."""

    def __init__(self, device=0):
        self.dll = load_library()
        self.ctx = C.c_void_p()
        rc = self.dll.mad_init(C.c_int(device), C.byref(self.ctx))
        if rc != 0:
            msg = self.dll.mad_last_error(None)
            self.ctx = None
            raise MadBackendError("MaD> mad_init(device=%d) failed (%s): %s -- the hot path runs on an MI355X only"
                                  % (device, ERRORS.get(rc, rc), msg.decode() if msg else ""))
        self.device = device
        self._fields = {}      # id(array) -> slot bookkeeping is done by the callers
        self._next_slot = 0
        self._free_slots = []
        self._eq_loaded = {}

    # -- plumbing -----------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self.dll.mad_last_error(self.ctx)
            raise MadBackendError("MaD> %s: %s" % (ERRORS.get(rc, rc), msg.decode() if msg else ""))

    def close(self):
        if self.ctx:
            self.dll.mad_destroy(self.ctx)
        self.ctx = None

    def synchronize(self):
        self._chk(self.dll.mad_synchronize(self.ctx))

    def stream(self):
        return self.dll.mad_stream(self.ctx)

    def set_overlap(self, on=True):
        """False: kernels of all lanes run one at a time (per-kernel timing); True (default): lanes overlap."""
        self._chk(self.dll.mad_set_overlap(self.ctx, C.c_int(1 if on else 0)))

    def timing_enable(self, on=True):
        self._chk(self.dll.mad_timing_enable(self.ctx, C.c_int(1 if on else 0)))

    def timing_reset(self):
        self._chk(self.dll.mad_timing_reset(self.ctx))

    def set_batching(self, on):
        """One GEMM launch for all matches of a `match_topk_many` bracket (True) or one per match (False, default)."""
        self._chk(self.dll.mad_set_batching(self.ctx, C.c_int(1 if on else 0)))

    def set_option(self, name, value):
        """Tuning values that change speed, never results (`mad_set_option`)."""
        self._chk(self.dll.mad_set_option(self.ctx, name.encode(), C.c_double(float(value))))

    def probe_peaks(self):
        """(device copy GB/s, int8 MFMA TOP/s) measured now on this device (mad_probe_peaks)."""
        c, m = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.dll.mad_probe_peaks(self.ctx, C.byref(c), C.byref(m)))
        return c.value, m.value

    def timing_get(self, what):
        t = C.c_double(0)
        n = C.c_int64(0)
        self._chk(self.dll.mad_timing_get(self.ctx, what.encode(), C.byref(t), C.byref(n)))
        return t.value, n.value

    # -- tables and fields ----------------------------------------------------------
    def set_eqsp(self, which, bounds, to_dom=None, adj_sec=None):
        bounds = _c(bounds, np.float64)
        to_dom = None if to_dom is None else _c(to_dom, np.float64)
        adj_sec = None if adj_sec is None else _c(adj_sec, np.float64)
        self._chk(self.dll.mad_set_eqsp(self.ctx, C.c_int(which), C.c_int(len(bounds)), _p(bounds), _p(to_dom), _p(adj_sec)))

    def set_orient_window(self, gw_sig=0.0):
        """Orientator(gw_sig): Gaussian window on the orientation histogram (0 = none) for the following `orient` calls.  The window is
        state of the context; `set_build`, `set_build_many` and `BuildBatch.run` set it themselves for every build (their `gw_sig`,
        default 0), so what is set here does not reach them.  A change of window waits for the device (mad_synchronize)."""
        if getattr(self, "_gw_sig", 0.0) != float(gw_sig):
            self._chk(self.dll.mad_set_orient_window(self.ctx, C.c_double(float(gw_sig))))
            self._gw_sig = float(gw_sig)

    def new_slot(self):
        if self._free_slots:
            return self._free_slots.pop()
        s = self._next_slot
        self._next_slot += 1
        if s >= 64:
            raise MadBackendError("MaD> out of gradient-field slots; free some with free_field()")
        return s

    def upload_field(self, slot, grad):
        """grad: the reference's grad_list entry, shape (X, Y, Z, 3), any strides; or a (3, X, Y, Z) array."""
        g = np.asarray(grad)
        if g.ndim != 4:
            raise ValueError("gradient field must be 4-D")
        if g.shape[-1] == 3:
            g = np.moveaxis(g, -1, 0)
        elif g.shape[0] != 3:
            raise ValueError("gradient field must be (X, Y, Z, 3) or (3, X, Y, Z)")
        planes = [_c(g[i], np.float32) for i in range(3)]
        nx, ny, nz = planes[0].shape
        self._chk(self.dll.mad_upload_field(self.ctx, C.c_int(slot), _p(planes[0]), _p(planes[1]), _p(planes[2]),
                                            C.c_int(nx), C.c_int(ny), C.c_int(nz)))

    def upload_field_device(self, slot, dev_ptr, nx, ny, nz):
        self._chk(self.dll.mad_upload_field_device(self.ctx, C.c_int(slot), C.c_void_p(dev_ptr), C.c_int(nx), C.c_int(ny), C.c_int(nz)))

    def free_field(self, slot):
        self._chk(self.dll.mad_free_field(self.ctx, C.c_int(slot)))
        if slot not in self._free_slots:
            self._free_slots.append(slot)

    # -- stage API --------------------------------------------------------------------
    def orient(self, slot, octave, coords, r=8, lim_main=6, lim_sec=6, want_counts=True, Z=112):
        coords = _c(coords, np.int32).reshape(-1, 3)
        n = len(coords)
        cap = max(1, n * lim_main * lim_sec)
        ra, rm, rs = (np.zeros(cap, np.int32) for _ in range(3))
        R = np.zeros((cap, 9))
        cnt = np.zeros((cap, Z), np.int32) if want_counts else None
        nrows = C.c_int64(0)
        nrej = C.c_int32(0)
        self._chk(self.dll.mad_orient(self.ctx, C.c_int(slot), C.c_int(octave), _p(coords), C.c_int(n), C.c_int(r),
                                      C.c_int(lim_main), C.c_int(lim_sec), _p(ra), _p(rm), _p(rs), _p(R), _p(cnt),
                                      C.byref(nrows), C.c_int64(cap), C.byref(nrej)))
        k = nrows.value
        return dict(anchor=ra[:k].copy(), main=rm[:k].copy(), sec=rs[:k].copy(), R=R[:k].reshape(k, 3, 3).copy(),
                    counts=None if cnt is None else cnt[:k].copy(), n_reject=nrej.value)

    def describe(self, slot, octave, coords, R, r=8, Zd=16, dsc_size=64):
        coords = _c(coords, np.int32).reshape(-1, 3)
        R = _c(R, np.float64).reshape(-1, 9)
        n = len(coords)
        out = np.zeros((n, dsc_size * Zd), np.int16)
        self._chk(self.dll.mad_describe_sized(self.ctx, C.c_int(slot), C.c_int(octave), _p(coords), _p(R), C.c_int64(n), C.c_int(r),
                                              C.c_int(dsc_size), _p(out)))
        return out

    def correlate(self, hi, lo, cc):
        hi, lo = _pad_rows(_c(hi, np.int16)), _pad_rows(_c(lo, np.int16))
        n_hi, D = hi.shape
        n_lo = lo.shape[0]
        npairs = C.c_int64(0)
        cap = 1 << 16
        while True:
            ph, pl, ps = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
            rc = self.dll.mad_correlate(self.ctx, _p(hi), C.c_int64(n_hi), _p(lo), C.c_int64(n_lo), C.c_int(D), C.c_double(cc),
                                        _p(ph), _p(pl), _p(ps), C.byref(npairs), C.c_int64(cap))
            if rc == -28:      # ENOSPC: the needed size came back in npairs
                cap = npairs.value
                continue
            self._chk(rc)
            break
        k = npairs.value
        return ph[:k].copy(), pl[:k].copy(), ps[:k].copy()

    def pose_score(self, pair_hi, pair_lo, pair_score, hi_p, hi_R, hi_meta, lo_p, lo_R, lo_meta, hi_cloud, lo_cloud,
                   dist=4.0, want_results=True):
        pair_hi, pair_lo = _c(pair_hi, np.int32), _c(pair_lo, np.int32)
        pair_score = _c(pair_score, np.float64)
        hi_p, lo_p = _c(hi_p, np.float64), _c(lo_p, np.float64)
        hi_R, lo_R = _c(hi_R, np.float64).reshape(-1, 9), _c(lo_R, np.float64).reshape(-1, 9)
        hi_meta, lo_meta = _c(hi_meta, np.int32), _c(lo_meta, np.int32)
        hi_cloud, lo_cloud = _c(hi_cloud, np.float64), _c(lo_cloud, np.float64)
        n = len(pair_hi)
        res = np.zeros((n, RESULT_COLS)) if want_results else None
        cnt = np.zeros(n, np.int32)
        self._chk(self.dll.mad_pose_score(self.ctx, _p(pair_hi), _p(pair_lo), _p(pair_score), C.c_int64(n),
                                          _p(hi_p), _p(hi_R), _p(hi_meta), C.c_int64(len(hi_p)),
                                          _p(lo_p), _p(lo_R), _p(lo_meta), C.c_int64(len(lo_p)),
                                          _p(hi_cloud), C.c_int64(len(hi_cloud)), _p(lo_cloud), C.c_int64(len(lo_cloud)),
                                          C.c_double(dist), _p(res), _p(cnt)))
        return res, cnt

    def topk(self, counts, k):
        counts = _c(counts, np.int32)
        k = min(int(k), len(counts))
        order = np.zeros(max(k, 1), np.int64)
        self._chk(self.dll.mad_topk(self.ctx, _p(counts), C.c_int64(len(counts)), C.c_int64(k), _p(order)))
        return order[:k]

    # -- device-resident pipeline -------------------------------------------------------
    def set_build(self, slots, anc_coords, anc_octave, anc_subv, anc_index, r=8, lim_main=6, lim_sec=6, into=None, gw_sig=0.0):
        """Orient + describe the anchors into a device-resident set.  Asynchronous; pass `into` to rebuild an
        existing set in place (its device buffers are reused).  `gw_sig`: the Gaussian window of Orientator(gw_sig) for THIS
        build (the window is state of the context: it is set here for every build, so that an Orientator object used earlier on
        the same context cannot leak its window into a set)."""
        self.set_orient_window(gw_sig)
        s = into if into is not None else DeviceSet(self)
        slots = (C.c_int * 2)(int(slots[0]), int(slots[1]))
        anc_coords = _c(anc_coords, np.int32).reshape(-1, 3)
        anc_octave = _c(anc_octave, np.int32)
        anc_subv = _c(anc_subv, np.float64).reshape(-1, 3)
        anc_index = _c(anc_index, np.int32)
        s.n_anchors = len(anc_octave)
        self._chk(self.dll.mad_set_build(self.ctx, s.h, slots, _p(anc_coords), _p(anc_octave), _p(anc_subv), _p(anc_index),
                                         C.c_int(len(anc_octave)), C.c_int(r), C.c_int(lim_main), C.c_int(lim_sec)))
        return s

    def prepare_build_many(self, jobs, r=8, lim_main=6, lim_sec=6, gw_sig=0.0):
        """jobs: [(slots, anc_coords, anc_octave, anc_subv, anc_index, into-or-None), ...] -> a BuildBatch whose `run()` enqueues
        orientation + description of all of them with one launch per stage (`mad_set_build_many`) and returns the sets.
        Prepared once, run every step: the argument arrays are converted and pinned down here."""
        return BuildBatch(self, jobs, r, lim_main, lim_sec, gw_sig)

    def set_build_many(self, jobs, r=8, lim_main=6, lim_sec=6, gw_sig=0.0):
        return self.prepare_build_many(jobs, r, lim_main, lim_sec, gw_sig).run()

    def set_load(self, row_anchor, row_main, row_R, dsc, anc_subv, anc_index, anc_octave):
        s = DeviceSet(self)
        row_anchor, row_main = _c(row_anchor, np.int32), _c(row_main, np.int32)
        row_R = _c(row_R, np.float64).reshape(-1, 9)
        dsc = _c(dsc, np.int16)
        if dsc.ndim != 2:
            dsc = dsc.reshape(len(row_anchor), -1)
        dsc = _pad_rows(dsc)
        anc_subv = _c(anc_subv, np.float64).reshape(-1, 3)
        anc_index, anc_octave = _c(anc_index, np.int32), _c(anc_octave, np.int32)
        s.n_anchors = len(anc_index)
        self._chk(self.dll.mad_set_load(self.ctx, s.h, C.c_int64(len(row_anchor)), _p(row_anchor), _p(row_main), _p(row_R),
                                        _p(dsc), C.c_int(dsc.shape[1] if len(dsc) else 1024), _p(anc_subv), _p(anc_index),
                                        _p(anc_octave), C.c_int(len(anc_index))))
        return s

    def match_topk(self, hi, lo, cc, dist, k):
        k = int(k)
        res = np.zeros((max(k, 1), RESULT_COLS))
        idx = np.zeros(max(k, 1), np.int64)
        n_out = C.c_int64(0)
        stats = np.zeros(4, np.int64)
        self._chk(self.dll.mad_match_topk(self.ctx, hi.h, lo.h, C.c_double(cc), C.c_double(dist), C.c_int64(k), _p(res), _p(idx),
                                          C.byref(n_out), _p(stats)))
        g = n_out.value
        return res[:g], idx[:g], dict(n_pairs=int(stats[0]), l_hi=int(stats[1]), l_lo=int(stats[2]), n_corr=int(stats[3]))

    def match_topk_many(self, his, lo, cc, dist, k):
        """[(rows, pair_index, stats)] for every subunit set of `his` against `lo`; see mad_match_topk_many."""
        k = int(k)
        n = len(his)
        res = np.zeros((max(n, 1), max(k, 1), RESULT_COLS))
        idx = np.zeros((max(n, 1), max(k, 1)), np.int64)
        n_out = np.zeros(max(n, 1), np.int64)
        stats = np.zeros((max(n, 1), 4), np.int64)
        arr = (C.c_void_p * max(n, 1))(*[h.h.value for h in his])
        self._chk(self.dll.mad_match_topk_many(self.ctx, C.c_int(n), arr, lo.h, C.c_double(cc), C.c_double(dist), C.c_int64(k),
                                               _p(res), _p(idx), _p(n_out), _p(stats)))
        out = []
        for i in range(n):
            g = int(n_out[i])
            out.append((res[i, :g], idx[i, :g], dict(n_pairs=int(stats[i, 0]), l_hi=int(stats[i, 1]), l_lo=int(stats[i, 2]),
                                                     n_corr=int(stats[i, 3]))))
        return out

    def last_pose_kernel(self):
        """0 k_pose_lds, 1 k_pose_lds32, 2 k_pose (global cell list): the kernel of the most recently enqueued match."""
        return int(self.dll.mad_last_pose_kernel(self.ctx))

    def last_pose_selected(self):
        """Pairs of the last completed match that went through the exact pose search (the rest were excluded by their bounds)."""
        return int(self.dll.mad_last_pose_selected(self.ctx))

    def device_allocations(self):
        """Device buffers (re)allocated by this context so far (mad_device_allocations): 0 new ones across a steady-state region."""
        return int(self.dll.mad_device_allocations(self.ctx))

    def match_topk_many_begin(self, his, lo, cc, dist, k):
        """Enqueue every match and return a handle; `match_topk_many_finish(handle)` waits and unpacks.  In between
        the caller may build the sets of its next batch (not the ones this bracket reads)."""
        k = int(k)
        n = len(his)
        h = dict(n=n, k=k, res=np.zeros((max(n, 1), max(k, 1), RESULT_COLS)), idx=np.zeros((max(n, 1), max(k, 1)), np.int64),
                 n_out=np.zeros(max(n, 1), np.int64), stats=np.zeros((max(n, 1), 4), np.int64), sets=(list(his), lo))
        arr = (C.c_void_p * max(n, 1))(*[x.h.value for x in his])
        self._chk(self.dll.mad_match_topk_many_begin(self.ctx, C.c_int(n), arr, lo.h, C.c_double(cc), C.c_double(dist), C.c_int64(k),
                                                     _p(h["res"]), _p(h["idx"]), _p(h["n_out"]), _p(h["stats"])))
        self._open_brackets = getattr(self, "_open_brackets", []) + [h]      # at most three (MAD_BRACKETS); they finish in the order they began
        return h

    def match_topk_many_finish(self, h):
        pending = getattr(self, "_open_brackets", [])
        if not pending or pending[0] is not h:
            raise MadBackendError("MaD> match_topk_many_finish: brackets finish in the order they were begun")
        self._open_brackets = pending[1:]      # the library closes the bracket whether or not one of its matches failed
        self._chk(self.dll.mad_match_topk_many_finish(self.ctx))
        out = []
        for i in range(h["n"]):
            g = int(h["n_out"][i])
            st = h["stats"][i]
            out.append((h["res"][i, :g], h["idx"][i, :g], dict(n_pairs=int(st[0]), l_hi=int(st[1]), l_lo=int(st[2]), n_corr=int(st[3]))))
        return out

    def match_shard_pairs(self, hi, lo, lo_begin, lo_end, cc):
        """Stage B of a sharded match -> (used_hi flags, used_lo flags, n_pairs) of the lo-row block [lo_begin, lo_end)."""
        used_hi, used_lo = np.zeros(max(hi.n_anchors, 1), np.uint8), np.zeros(max(lo.n_anchors, 1), np.uint8)
        n = C.c_int64(0)
        self._chk(self.dll.mad_match_shard_pairs(self.ctx, hi.h, lo.h, C.c_int64(lo_begin), C.c_int64(lo_end), C.c_double(cc),
                                                 _p(used_hi), _p(used_lo), C.byref(n)))
        return used_hi[:hi.n_anchors], used_lo[:lo.n_anchors], n.value

    def match_shard_topk(self, hi, lo, used_hi_all, used_lo_all, dist, k):
        """Stage C -> (rows (m, 23), counts (m,), global pair ranks (m,), l_hi) with m <= k."""
        uh, ul = _c(used_hi_all, np.uint8), _c(used_lo_all, np.uint8)
        assert len(uh) == hi.n_anchors and len(ul) == lo.n_anchors
        res = np.zeros((k, RESULT_COLS))
        rank, cnt = np.zeros(k, np.int64), np.zeros(k, np.int32)
        n, l_hi = C.c_int64(0), C.c_int64(0)
        self._chk(self.dll.mad_match_shard_topk(self.ctx, hi.h, lo.h, _p(uh), _p(ul), C.c_double(dist), C.c_int64(k), _p(res), _p(rank),
                                                _p(cnt), C.byref(n), C.byref(l_hi)))
        m = n.value
        return res[:m].copy(), cnt[:m].copy(), rank[:m].copy(), l_hi.value

    def match_shard_record_doubles(self, k):
        return int(self.dll.mad_match_shard_record_doubles(C.c_int64(int(k))))

    def match_shard_begin(self, hi, lo, lo_begin, lo_end, n_lo, cc, flags_ptr):
        """Stage B, asynchronous on hi's lane: the shard's flags go to device memory at `flags_ptr` (hi.n_anchors + lo.n_anchors bytes)."""
        self._chk(self.dll.mad_match_shard_begin(self.ctx, hi.h, lo.h, C.c_int64(int(lo_begin)), C.c_int64(int(lo_end)), C.c_int64(int(n_lo)),
                                                 C.c_double(cc), C.c_void_p(int(flags_ptr))))

    def match_shard_score(self, hi, lo, flags_all_ptr, dist, k, out_ptr):
        """Stage C, asynchronous: the shard's record (match_shard_record_doubles(k) float64) goes to device memory at `out_ptr`."""
        self._chk(self.dll.mad_match_shard_score(self.ctx, hi.h, lo.h, C.c_void_p(int(flags_all_ptr)), C.c_double(dist), C.c_int64(int(k)),
                                                 C.c_void_p(int(out_ptr))))

    def match_shard_collect(self, hi, all_ptr, n):
        """The n float64 at device address `all_ptr` on their way to the host, behind everything enqueued on hi's lane -> ticket."""
        t = C.c_int(0)
        self._chk(self.dll.mad_match_shard_collect(self.ctx, hi.h, C.c_void_p(int(all_ptr)), C.c_int64(int(n)), C.byref(t)))
        return t.value

    def match_shard_wait(self, ticket, n):
        out = np.zeros(int(n))
        self._chk(self.dll.mad_match_shard_wait(self.ctx, C.c_int(int(ticket)), _p(out), C.c_int64(int(n))))
        return out

    # -- a structure's rows built in shares (SURVEY.md 8(e) stage A; the all-gather itself is mad_amd/dist.py's) --------
    def set_wire_bytes(self, cap_rows, D=1024):
        n = int(self.dll.mad_set_wire_bytes(C.c_int(int(D)), C.c_int64(int(cap_rows))))
        if n < 0:
            raise ValueError("set_wire_bytes(%r, %r)" % (cap_rows, D))
        return n

    def set_export(self, share, cap_rows, wire=None, device_ptr=None):
        """Wire image of a built set.  device_ptr: address of device memory of set_wire_bytes(cap_rows) bytes (asynchronous
        on the set's stream); otherwise a host uint8 array is filled (and returned)."""
        if device_ptr is not None:
            self._chk(self.dll.mad_set_export(self.ctx, share.h, C.c_void_p(int(device_ptr)), C.c_int(1), C.c_int64(int(cap_rows))))
            return None
        nbytes = self.set_wire_bytes(cap_rows)
        if wire is None:
            wire = np.zeros(nbytes, np.uint8)
        if wire.dtype != np.uint8 or wire.size != nbytes or not wire.flags.c_contiguous:
            raise ValueError("set_export: wire must be a contiguous uint8 array of %d bytes" % nbytes)
        self._chk(self.dll.mad_set_export(self.ctx, share.h, _p(wire), C.c_int(0), C.c_int64(int(cap_rows))))
        return wire

    def set_import(self, n_shares, cap_rows, anc_coords, anc_octave, anc_subv, anc_index, wires=None, device_ptr=None, into=None):
        """The full set of a structure from the wire images of its n_shares shares (anchor a was built in share a % n_shares).
        wires: host uint8 array (n_shares x set_wire_bytes), or device_ptr: the same bytes in device memory (asynchronous)."""
        s = into if into is not None else DeviceSet(self)
        anc_octave = _c(anc_octave, np.int32)
        anc_subv = _c(anc_subv, np.float64).reshape(-1, 3)
        anc_index = _c(anc_index, np.int32)
        anc_coords = None if anc_coords is None else _c(anc_coords, np.int32).reshape(-1, 3)
        n = len(anc_octave)
        if len(anc_subv) != n or len(anc_index) != n or (anc_coords is not None and len(anc_coords) != n):
            raise ValueError("set_import: anchor arrays of different lengths")
        if device_ptr is not None:
            src, on_dev = C.c_void_p(int(device_ptr)), 1
        else:
            wires = np.ascontiguousarray(wires, dtype=np.uint8)
            if wires.size != n_shares * self.set_wire_bytes(cap_rows):
                raise ValueError("set_import: %d bytes for %d shares of %d" % (wires.size, n_shares, self.set_wire_bytes(cap_rows)))
            src, on_dev = _p(wires), 0
        s.n_anchors = n
        self._chk(self.dll.mad_set_import(self.ctx, s.h, src, C.c_int(on_dev), C.c_int(int(n_shares)), C.c_int64(int(cap_rows)),
                                          _p(anc_coords), _p(anc_octave), _p(anc_subv), _p(anc_index), C.c_int(n)))
        return s

    def match_fetch(self, n_pairs):
        ph, pl = np.zeros(n_pairs, np.int32), np.zeros(n_pairs, np.int32)
        ps, cn = np.zeros(n_pairs), np.zeros(n_pairs, np.int32)
        self._chk(self.dll.mad_match_fetch(self.ctx, _p(ph), _p(pl), _p(ps), _p(cn), C.c_int64(n_pairs)))
        return ph, pl, ps, cn

    def match_results(self, hi, lo, n_pairs):
        res = np.zeros((max(n_pairs, 1), RESULT_COLS))
        self._chk(self.dll.mad_match_results(self.ctx, hi.h, lo.h, _p(res), C.c_int64(n_pairs)))
        return res[:n_pairs]

    def match_used(self, n_hi_anchors, n_lo_anchors):
        uh, ul = np.zeros(max(n_hi_anchors, 1), np.uint8), np.zeros(max(n_lo_anchors, 1), np.uint8)
        self._chk(self.dll.mad_match_used(self.ctx, _p(uh), C.c_int32(n_hi_anchors), _p(ul), C.c_int32(n_lo_anchors)))
        return uh[:n_hi_anchors].astype(bool), ul[:n_lo_anchors].astype(bool)

    # -- refinement / density / ccc --------------------------------------------------------
    def upload_density(self, grid, origin, voxsp):
        grid = _c(grid, np.float32)
        nx, ny, nz = grid.shape
        self._chk(self.dll.mad_upload_density(self.ctx, _p(grid), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                                              C.c_double(origin[0]), C.c_double(origin[1]), C.c_double(origin[2]), C.c_double(voxsp)))

    def refine(self, coords, n_steps=500, max_step=0.5, min_step=0.01):
        """coords: (n_cand, n_atoms, 3) or (n_atoms, 3).  Returns (coords, converged[], last_step[])."""
        c = _c(coords, np.float64).copy()
        single = c.ndim == 2
        if single:
            c = c[None]
        n_cand, n_atoms, _ = c.shape
        conv, last = np.zeros(n_cand, np.int32), np.zeros(n_cand, np.int32)
        self._chk(self.dll.mad_refine(self.ctx, _p(c), C.c_int(n_cand), C.c_int64(n_atoms), C.c_int(n_steps),
                                      C.c_double(max_step), C.c_double(min_step), _p(conv), _p(last)))
        if single:
            return c[0], bool(conv[0]), int(last[0])
        return c, conv.astype(bool), last

    def structure_to_density(self, atoms, mass, resolution, voxsp, isovalue=0.0, pad=0):
        atoms, mass = _c(atoms, np.float64), _c(mass, np.float64)
        dims = np.zeros(3, np.int32)
        org = np.zeros(3)
        args = (self.ctx, _p(atoms), _p(mass), C.c_int64(len(atoms)), C.c_double(resolution), C.c_double(voxsp),
                C.c_double(isovalue), C.c_int(pad), _p(dims), _p(org))
        self._chk(self.dll.mad_structure_to_density(*args, None))
        grid = np.zeros(tuple(int(d) for d in dims), np.float32)
        self._chk(self.dll.mad_structure_to_density(*args, _p(grid)))
        return grid, float(org[0]), float(org[1]), float(org[2])

    def density_ccc(self, coords, mass, resolution, density_isovalue=0.0, ccc_isovalue=0.0):
        """coords: (n_cand, n_atoms, 3) placed copies of one structure -> CCC of each copy's simulated density with the
        map given to upload_density (PDB.structure_to_density + Dmap.get_CCC_with_grid, on the device end to end)."""
        c = _c(coords, np.float64)
        if c.ndim == 2:
            c = c[None]
        n_cand, n_atoms, _ = c.shape
        mass = _c(mass, np.float64)
        out = np.zeros(n_cand, np.float64)
        self._chk(self.dll.mad_density_ccc(self.ctx, _p(c), _p(mass), C.c_int(n_cand), C.c_int64(n_atoms), C.c_double(resolution),
                                           C.c_double(density_isovalue), C.c_double(ccc_isovalue), _p(out)))
        return out

    def dock_refine_score(self, base_atoms, mass, hi_p, lo_p, rot, resolution, n_steps=500, max_step=0.5, min_step=0.01,
                          density_isovalue=0.0, ccc_isovalue=0.0, want_coords=True, cand_struct=None):
        """Candidate poses -> (refined coords or None, converged, last_step, ccc), device-resident from the poses to the scores
        (mad_dock_refine_score): start_c = (atoms - hi_p[c]) @ rot[c] + lo_p[c], refined against the uploaded map, turned into a
        simulated density and scored by CCC.  One structure: base_atoms (n, 3), mass (n) -> coords (n_cand, n, 3).  Several: lists of
        them + cand_struct[c] = the structure candidate c is a pose of -> coords = list of (n_of_c, 3) arrays."""
        many = isinstance(base_atoms, (list, tuple))
        bases = [_c(a, np.float64).reshape(-1, 3) for a in (base_atoms if many else [base_atoms])]
        masses = [_c(m, np.float64).reshape(-1) for m in (mass if many else [mass])]
        assert len(bases) == len(masses) and all(len(a) == len(m) for a, m in zip(bases, masses))
        first = np.zeros(len(bases) + 1, np.int64)
        first[1:] = np.cumsum([len(a) for a in bases])
        base_all, mass_all = np.concatenate(bases), np.concatenate(masses)
        hi_p, lo_p = _c(hi_p, np.float64).reshape(-1, 3), _c(lo_p, np.float64).reshape(-1, 3)
        rot = _c(rot, np.float64).reshape(-1, 9)
        n_cand = len(hi_p)
        cs = np.zeros(n_cand, np.int32) if cand_struct is None else _c(cand_struct, np.int32).reshape(-1)
        assert len(lo_p) == n_cand and len(rot) == n_cand and len(cs) == n_cand
        sizes = (first[1:] - first[:-1])[cs] if n_cand else np.zeros(0, np.int64)
        flat = np.zeros((int(sizes.sum()), 3)) if want_coords else None
        conv, last = np.zeros(max(n_cand, 1), np.int32), np.zeros(max(n_cand, 1), np.int32)
        ccc = np.zeros(max(n_cand, 1))
        self._chk(self.dll.mad_dock_refine_score(self.ctx, C.c_int(len(bases)), _p(base_all), _p(mass_all), _p(first), C.c_int(n_cand), _p(cs),
                                                 _p(hi_p), _p(lo_p), _p(rot), C.c_int(int(n_steps)), C.c_double(max_step), C.c_double(min_step),
                                                 C.c_double(resolution), C.c_double(density_isovalue), C.c_double(ccc_isovalue),
                                                 _p(flat) if want_coords else None, _p(conv), _p(last), _p(ccc)))
        coords = None
        if want_coords:
            if many:
                ends = np.cumsum(sizes)
                coords = [flat[e - n:e] for e, n in zip(ends, sizes)]
            else:
                coords = flat.reshape(n_cand, len(bases[0]), 3)
        return coords, conv[:n_cand].astype(bool), last[:n_cand], ccc[:n_cand]

    def grid_overlap(self, g1, o1, g2, o2, voxsp, isovalue=1e-8):
        """-> (common, positives of g1); both grids (writable C-contiguous float32) are clamped in place."""
        for g in (g1, g2):
            if g.dtype != np.float32 or not g.flags.c_contiguous or not g.flags.writeable:
                raise ValueError("grid_overlap needs writable C-contiguous float32 grids")
        d1, d2 = np.array(g1.shape, np.int32), np.array(g2.shape, np.int32)
        o1, o2 = _c(o1, np.float64), _c(o2, np.float64)
        common, npos = C.c_int64(0), C.c_int64(0)
        self._chk(self.dll.mad_grid_overlap(self.ctx, _p(g1), _p(d1), _p(o1), _p(g2), _p(d2), _p(o2), C.c_double(voxsp),
                                            C.c_double(isovalue), C.byref(common), C.byref(npos)))
        return common.value, npos.value

    def overlap_matrix(self, coords_list, mass_list, resolution=5.0, voxsp=2.0, density_isovalue=0.2, overlap_isovalue=1e-8):
        """Upper-triangular table of get_overlap between the simulated densities of the given structures."""
        n = len(coords_list)
        out = np.zeros((n, n), np.float64)
        if n < 2:
            return out
        first = np.zeros(n + 1, np.int64)
        first[1:] = np.cumsum([len(c) for c in coords_list])
        atoms = _c(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in coords_list]), np.float64)
        mass = _c(np.concatenate([np.asarray(m, np.float64).reshape(-1) for m in mass_list]), np.float64)
        if len(mass) != len(atoms):
            raise ValueError("overlap_matrix: %d masses for %d atoms" % (len(mass), len(atoms)))
        self._chk(self.dll.mad_overlap_matrix(self.ctx, _p(atoms), _p(mass), _p(first), C.c_int(n), C.c_double(resolution),
                                              C.c_double(voxsp), C.c_double(density_isovalue), C.c_double(overlap_isovalue), _p(out)))
        return out

    def ccc(self, g1, o1, g2, o2, voxsp, isovalue=0.0):
        """Both grids must be C-contiguous float32; they are clamped in place like the reference."""
        for g in (g1, g2):
            if g.dtype != np.float32 or not g.flags.c_contiguous or not g.flags.writeable:
                raise ValueError("ccc needs writable C-contiguous float32 grids")
        d1, d2 = np.array(g1.shape, np.int32), np.array(g2.shape, np.int32)
        o1, o2 = _c(o1, np.float64), _c(o2, np.float64)
        out = C.c_double(0)
        self._chk(self.dll.mad_ccc(self.ctx, _p(g1), _p(d1), _p(o1), _p(g2), _p(d2), _p(o2), C.c_double(voxsp),
                                   C.c_double(isovalue), C.byref(out)))
        return out.value


_default = None


def get_lib(device=None):
    """Process-wide default context (device from MAD_DEVICE / LOCAL_RANK / 0)."""
    global _default
    if _default is None:
        if device is None:
            device = int(os.environ.get("MAD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default = Lib(device)
    return _default


def reset_lib():
    global _default
    if _default is not None:
        _default.close()
    _default = None
