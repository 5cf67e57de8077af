"""Device scale space (mad_space_build / mad_space_peaks) against the scipy restatement of the reference's
MapSpace (oracle/scale_space.py) and against the reference's own output (tests/golden/g_mapspace.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mad_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _density(lib, n_atoms, radius, seed, res=8.0, vs=1.6):
    atoms, names, elems = synth.random_globule(n_atoms, radius, seed=seed)
    grid, x0, y0, z0 = lib.structure_to_density(atoms, synth.masses(elems), res, vs)
    return grid


def _ulp_report(a, b, floor=1e-12):
    """(number of differing voxels, largest difference in float32 ulps of the reference value).  Differences
    below `floor` (the grids are normalised to a maximum of 1) are rounding noise of the float64 spline in the
    zero-padded rim, where the values themselves are ~1e-17, and are not counted."""
    diff = np.abs(a.astype(np.float64) - b.astype(np.float64)) > floor
    if not diff.any():
        return 0, 0.0
    ulp = np.spacing(np.abs(b[diff]).astype(np.float32)).astype(np.float64)
    return int(diff.sum()), float(np.max(np.abs(a[diff].astype(np.float64) - b[diff].astype(np.float64)) / ulp))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_volumes_match_scipy(lib, dtype):
    from mad_amd._lib import DeviceSpace
    from oracle import scale_space as OS
    grid = _density(lib, 1500, 16.0, seed=3).astype(dtype)
    if dtype is np.float64:
        grid = grid / np.amax(grid).astype(np.float32)
    ref = OS.build_volumes(grid, pad=9, oct_mode="both", sig_init=2, sig_presmooth=1)
    sp = DeviceSpace(lib).build(grid, pad=9, oct_mode="both", sig_init=2, sig_presmooth=1)
    assert sp.kinds == [0, 1]
    assert [tuple(g.shape) for g in ref["grid_list"]] == sp.shapes
    assert sp.dtypes == [np.float32, dtype]
    # base octave: every pass follows scipy's summation order and rounding -> identical bits
    np.testing.assert_array_equal(sp.download(1, DeviceSpace.GRID), ref["grid_list"][1])
    np.testing.assert_array_equal(sp.download(1, DeviceSpace.GAUSS), ref["gauss_list"][1])
    np.testing.assert_array_equal(sp.download(1, DeviceSpace.LOG), ref["map_space"][1])
    # upsampled octave: the spline solve differs from LAPACK + de Boor by ~4e-16 relative before the float32
    # rounding, so a voxel in ~1e7 may land on the neighbouring float32; everything downstream is exact again
    up = sp.download(0, DeviceSpace.GRID)
    n_bad, worst = _ulp_report(up, ref["grid_list"][0])
    assert n_bad <= max(2, up.size // 100000) and worst <= 1.0, (n_bad, worst)
    for what, key in ((DeviceSpace.GAUSS, "gauss_list"), (DeviceSpace.LOG, "map_space")):
        got = sp.download(0, what)
        np.testing.assert_allclose(got, ref[key][0], rtol=0, atol=1e-6 if n_bad else 1e-12)
    sp.close()


@pytest.mark.parametrize("mode", ["base", "up"])
def test_single_octave_modes(lib, mode):
    from mad_amd._lib import DeviceSpace
    from oracle import scale_space as OS
    grid = _density(lib, 800, 12.0, seed=4)
    ref = OS.build_volumes(grid, pad=9, oct_mode=mode, sig_init=2, sig_presmooth=1)
    sp = DeviceSpace(lib).build(grid, pad=9, oct_mode=mode, sig_init=2, sig_presmooth=1)
    assert len(sp.shapes) == 1 and sp.kinds == [1 if mode == "base" else 0]
    np.testing.assert_allclose(sp.download(0, DeviceSpace.LOG), ref["map_space"][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sp.download(0, DeviceSpace.GAUSS), ref["gauss_list"][0], rtol=0, atol=1e-6)
    sp.close()


def test_gradient_texels_feed_the_kernels(lib):
    """The texels written by the build equal an upload of np.gradient(gauss): same descriptors from both slots."""
    from mad_amd._lib import DeviceSpace
    from mad_amd.eqsp import EQSP_Sphere
    lib.set_eqsp(1, EQSP_Sphere(16).sphere_eqsp)
    grid = _density(lib, 2500, 20.0, seed=6)
    s_up, s_base = lib.new_slot(), lib.new_slot()
    sp = DeviceSpace(lib).build(grid, slot_up=s_up, slot_base=s_base)
    rng = np.random.default_rng(0)
    for entry, slot in ((0, s_up), (1, s_base)):
        gauss = sp.download(entry, DeviceSpace.GAUSS)
        ref_slot = lib.new_slot()
        lib.upload_field(ref_slot, np.moveaxis(np.array(np.gradient(gauss)), 0, -1))
        shape = np.array(gauss.shape)
        margin = 28 if entry == 0 else 14
        coords = np.stack([rng.integers(margin, s - margin, 40) for s in shape], 1).astype(np.int32)
        R = np.array([synth.random_rotation(rng) for _ in range(len(coords))])
        a = lib.describe(slot, entry, coords, R)
        b = lib.describe(ref_slot, entry, coords, R)
        assert a.any()
        np.testing.assert_array_equal(a, b)
        lib.free_field(ref_slot)
    lib.free_field(s_up)
    lib.free_field(s_base)
    sp.close()


def test_peaks_match_the_maximum_filter(lib):
    from mad_amd._lib import DeviceSpace
    from oracle import scale_space as OS
    grid = _density(lib, 6000, 28.0, seed=7, res=6.0, vs=1.5)
    sp = DeviceSpace(lib).build(grid)
    for entry in (0, 1):
        log = sp.download(entry, DeviceSpace.LOG)
        ref = OS.peak_local_max(log, exclude_border=12, threshold_abs=5e-2)
        coords, vals = sp.peaks(entry, threshold=5e-2, border=12)
        assert len(ref) >= 4
        np.testing.assert_array_equal(coords, ref)
        np.testing.assert_array_equal(vals, log[tuple(ref.T)].astype(np.float64))
        patches = sp.patches(entry, coords[:7], 6)
        for c, p in zip(coords[:7], patches):
            np.testing.assert_array_equal(p, log[c[0] - 6:c[0] + 7, c[1] - 6:c[1] + 7, c[2] - 6:c[2] + 7])
    sp.close()


def test_mapspace_and_detector_against_reference(lib, tmp_path):
    """The drop-in classes on the reference's own map: samples of grad_list / map_space and the anchor list."""
    from mad_amd import _lib
    from mad_amd.Detector import Detector
    from mad_amd.MapSpace import MapSpace
    old, _lib._default = _lib._default, lib
    try:
        g = dict(np.load(os.path.join(GOLD, "g_mapspace.npz")))
        sit = str(tmp_path / "map.sit")
        open(sit, "w").write("x")      # only the extension is inspected before build_from_grid
        ms = MapSpace(sit, sig_init=2.0, sig_presmooth=1)
        ms.voxelsp = float(g["vs"])
        grid = g["map_grid"].astype(np.float64)
        grid = grid / np.amax(grid).astype(np.float32)      # what MapSpace.py:96 does to a situs map
        ms.build_from_grid(grid, *[float(v) for v in g["map_origin"]])
        np.testing.assert_allclose([ms.xi, ms.yi, ms.zi], g["origin"], atol=1e-12)
        for o in (0, 1):
            idx = g["grad_idx_%d" % o]
            assert tuple(g["grad_shape_%d" % o]) == ms.grad_list[o].shape
            assert ms.grad_list[o].dtype == g["grad_val_%d" % o].dtype
            # the fixture's map went through a 6-decimal situs text file; this one did not
            np.testing.assert_allclose(ms.grad_list[o][idx[:, 0], idx[:, 1], idx[:, 2]], g["grad_val_%d" % o], rtol=0, atol=2e-6)
            np.testing.assert_allclose(ms.map_space[o][idx[:, 0], idx[:, 1], idx[:, 2]], g["log_val_%d" % o], rtol=0, atol=2e-6)
        anchors = Detector().find_anchors(ms)
        # same anchors as at fixture time up to the text round-off of the map: same count, same voxels
        assert abs(len(anchors) - len(g["anchor_coords"])) <= 2
        got = {(a.oct_scale,) + tuple(int(v) for v in a.coords) for a in anchors}
        ref = {(int(o),) + tuple(int(v) for v in c) for o, c in zip(g["anchor_oct"], g["anchor_coords"])}
        assert len(got & ref) >= len(ref) - 2
        by_key = {(int(o),) + tuple(int(v) for v in c): s for o, c, s in zip(g["anchor_oct"], g["anchor_coords"], g["anchor_subv"])}
        for a in anchors:
            key = (a.oct_scale,) + tuple(int(v) for v in a.coords)
            if key in by_key:
                np.testing.assert_allclose(a.subv_map_coords, by_key[key], atol=1e-3)
        assert ms.device_slots(lib)[0] >= 0 and ms.device_slots(lib)[1] >= 0
        ms.release_device()
    finally:
        _lib._default = old
