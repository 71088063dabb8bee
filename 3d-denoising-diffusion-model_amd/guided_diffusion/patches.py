"""
Whole-body volume <-> 96^3 sub-volume patches: the steps either side of the
sampling hot path in the reference's inference script.

Restated from scripts/test.py:185-246 (tiling), :248-262 (3-D Hann window),
:92-146 (weighted overlap-add) and :283-301 (start positions).  scripts/test.py
cannot be imported in the build container (it needs tifffile / mpi4py) and the
reference holds no fixtures for it, so every function here is pinned to
outputs of the reference's OWN code, taken out of the script's syntax tree
and run unchanged (tests/golden/make_golden.py): the three pure helpers
(hann_window_3d, xy_starts, z_starts; script_helpers.npz) and the tiling /
overlap-add loop nests that are inline in load_data_for_worker and main()
(split_volume, stitch_patches; script_tiling.npz), bit for bit
(tests/test_patches_cpu.py).  One intended difference: voxels whose total
weight is 0 are 0 here, uninitialised memory in the reference (np.divide with
`where` and no `out`).

Host-side numpy on purpose: this is file-format glue around the GPU path (a
200x200x130 volume is 5 M voxels), exactly where the reference has it.
"""

import numpy as np


def xy_starts(dim_size, patch_size, num_patches=3):
    """scripts/test.py:283-293: fixed number of patches per axis."""
    if dim_size == 200 and patch_size == 96 and num_patches == 3:
        return [0, 52, 104]
    if num_patches == 1:
        return [0]
    step = (dim_size - patch_size) / (num_patches - 1)
    starts = [int(i * step) for i in range(num_patches)]
    starts[-1] = min(starts[-1], dim_size - patch_size)
    return starts


def z_starts(dim_size, patch_size):
    """scripts/test.py:295-301: one patch, or first + last with overlap."""
    if dim_size <= patch_size:
        return [0]
    return [0, dim_size - patch_size]


def patch_grid(shape_dhw, resolution, num_xy=3):
    """[(x_start, y_start, z_start)] in the reference's nesting order (x, y, z)."""
    D, H, W = shape_dhw
    return [(xs, ys, zs) for xs in xy_starts(H, resolution, num_xy)
            for ys in xy_starts(W, resolution, num_xy) for zs in z_starts(D, resolution)]


def split_volume(vol, resolution, num_xy=3):
    """(D,H,W) volume -> (P, 1, Z, H, W) float32 zero-padded patches + the grid.
    (The reference routes through an (H,W,Z) transpose and back; the result is the
    same (Z,H,W)-ordered patch.)"""
    vol = np.asarray(vol)
    if vol.ndim == 4 and vol.shape[0] == 1:
        vol = vol[0]
    if vol.ndim != 3:
        raise ValueError("expected a (D,H,W) volume, got shape %s" % (vol.shape,))
    vol = vol.astype(np.float32)
    D, H, W = vol.shape
    grid = patch_grid((D, H, W), resolution, num_xy)
    out = np.zeros((len(grid), 1, resolution, resolution, resolution), dtype=np.float32)
    for i, (xs, ys, zs) in enumerate(grid):
        p = vol[zs:zs + resolution, xs:xs + resolution, ys:ys + resolution]
        out[i, 0, :p.shape[0], :p.shape[1], :p.shape[2]] = p
    return out, grid


def hann_window_3d(size):
    """scripts/test.py:248-262: separable Hann window normalised to max 1."""
    h = np.hanning(size)
    w = np.outer(np.outer(h, h).ravel(), h).reshape(size, size, size)
    return w / w.max()


def stitch_patches(patches_hwz, grid, shape_dhw, resolution):
    """Weighted overlap-add of denoised patches (each (H,W,Z), the layout the
    reference permutes samples into, scripts/test.py:72) into an (H,W,Z) volume.
    Voxels whose total weight is 0 (the outermost planes: np.hanning is 0 at both
    ends) stay 0, as in the reference (np.divide(..., where=weight > 0))."""
    D, H, W = shape_dhw
    acc = np.zeros((H, W, D), dtype=np.float32)
    wsum = np.zeros_like(acc)
    win = hann_window_3d(resolution)
    for patch, (xs, ys, zs) in zip(patches_hwz, grid):
        patch = np.squeeze(np.asarray(patch))
        if patch.ndim != 3:
            raise ValueError("patch has unexpected dimensions: %s" % (patch.shape,))
        xe, ye, ze = min(xs + resolution, H), min(ys + resolution, W), min(zs + resolution, D)
        hx, wy, dz = xe - xs, ye - ys, ze - zs
        acc[xs:xe, ys:ye, zs:ze] += patch[:hx, :wy, :dz] * win[:hx, :wy, :dz]
        wsum[xs:xe, ys:ye, zs:ze] += win[:hx, :wy, :dz]
    return np.divide(acc, wsum, out=acc.copy(), where=wsum > 0), wsum


def load_volume(path):
    """Input volume as (D,H,W) float32.  .npz ('arr_0' or the first array) and .npy
    besides the reference's .tif/.tiff (tiff_io: tifffile when importable, else its own reader; scripts/test.py
    reads tif only, README.md:67 tells users to edit the loader for other formats)."""
    low = path.lower()
    if low.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            key = "arr_0" if "arr_0" in z.files else z.files[0]
            vol = z[key]
    elif low.endswith(".npy"):
        vol = np.load(path, allow_pickle=False)
    elif low.endswith((".tif", ".tiff")):
        from . import tiff_io
        vol = tiff_io.imread(path)
    else:
        raise ValueError("unsupported input file type: %s" % path)
    vol = np.asarray(vol)
    while vol.ndim > 3 and vol.shape[0] == 1:
        vol = vol[0]
    if vol.ndim != 3:
        raise ValueError("expected a (D,H,W) volume in %s, got %s" % (path, vol.shape))
    return vol.astype(np.float32)
