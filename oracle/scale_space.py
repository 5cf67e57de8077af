"""CPU restatement of the reference's scale-space stage -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(`mad_amd/`) never does.  It holds the device implementation (`mad_space_build`, `mad_space_peaks`) to
the arithmetic of the reference:

* `build_volumes` follows MapSpace.build_space, mad/MapSpace.py:116-189, with the very scipy calls the
  reference makes (interp1d cubic per axis, gaussian_filter, gaussian_laplace, np.gradient);
* `peak_local_max` restates skimage.feature.peak_local_max 0.17.2 as Detector.py:29 calls it (3x3x3 maximum
  filter with zero extension, strict threshold, border exclusion, descending intensity).  scikit-image is not
  installed: PARITY UNPINNED for this function.

Pinned by tests/golden/g_mapspace.npz (samples of the reference's own MapSpace output).
"""
import numpy as np
from scipy import ndimage as ndi
from scipy.interpolate import interp1d
from scipy.ndimage import gaussian_filter, gaussian_laplace


def build_volumes(grid, pad=9, oct_mode="both", sig_init=2, sig_presmooth=1):
    """-> dict(grid_list, map_space, gauss_list, grad_list); list entry 0 = upsampled octave when present."""
    if pad:
        grid = np.pad(grid, pad, mode="constant")
    xb, yb, zb = grid.shape
    grids = []
    if oct_mode in ("up", "both"):
        up = grid
        for axis, n in enumerate((xb, yb, zb)):      # MapSpace.interpn_so, :191-214
            up = interp1d(np.arange(0, n, 1), up, axis=axis, kind="cubic")(np.arange(0, n - 0.5, 0.5))
        if sig_presmooth:
            up = gaussian_filter(up, sigma=sig_presmooth)
        grids.append(up.astype(np.float32))
    if oct_mode in ("base", "both"):
        grids.append(grid)
    out = dict(grid_list=grids, map_space=[], gauss_list=[], grad_list=[])
    for g in grids:
        log_g = -1 * gaussian_laplace(g, sigma=sig_init) * sig_init ** 2
        log_g[log_g < 0] = 0.0
        out["map_space"].append(log_g)
        out["gauss_list"].append(gaussian_filter(g, sig_init))
        out["grad_list"].append(np.moveaxis(np.array(np.gradient(out["gauss_list"][-1])), 0, -1))
    return out


def peak_local_max(image, exclude_border=12, threshold_abs=5e-2, min_distance=1):
    size = 2 * min_distance + 1
    is_max = ndi.maximum_filter(image, size=size, mode="constant") == image
    is_max &= image > threshold_abs
    if exclude_border:
        b = int(exclude_border)
        for ax in range(image.ndim):
            sl = [slice(None)] * image.ndim
            sl[ax] = slice(None, b)
            is_max[tuple(sl)] = False
            sl[ax] = slice(-b, None)
            is_max[tuple(sl)] = False
    coords = np.transpose(np.nonzero(is_max))
    order = np.argsort(-image[tuple(coords.T)], kind="stable")
    return coords[order]
