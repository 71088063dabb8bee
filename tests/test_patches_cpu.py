"""
Patch tiler / Hann stitcher / volume I/O (guided_diffusion/patches.py).

The reference script (scripts/test.py) cannot be imported in the build container (tifffile / mpi4py)
and holds no fixtures.  Everything here is pinned to outputs of the reference's OWN code, lifted out
of the script's syntax tree and executed unchanged by tests/golden/make_golden.py: the three pure
helpers -- Hann window, the two start-position rules -- (script_helpers.npz) and the two loop nests
around them, the tiling of load_data_for_worker (:214-231) and the Hann overlap-add of main()
(:113-146) (script_tiling.npz).  The one intended difference: where the total weight is 0 the
reference's np.divide(..., where=...) leaves uninitialised memory, this package writes 0.
"""

import numpy as np
import pytest

from guided_diffusion import patches


def test_pure_helpers_vs_reference_golden(golden):
    g = golden("script_helpers.npz")
    for (dim, patch, n), ref in zip(g["xy_cases"], g["xy_starts"]):
        assert patches.xy_starts(int(dim), int(patch), int(n)) == [int(v) for v in ref[:n]], (dim, patch, n)
    for (dim, patch), ref in zip(g["z_cases"], g["z_starts"]):
        assert patches.z_starts(int(dim), int(patch)) == [int(v) for v in ref if v >= 0], (dim, patch)
    for size in (8, 16):
        assert np.array_equal(patches.hann_window_3d(size), g["hann%d" % size])      # same float operations
    w = patches.hann_window_3d(96)
    assert np.array_equal(w[48], g["hann96_mid_plane"])
    assert np.allclose([w.min(), w.max(), w.mean(), w.sum()], g["hann96_stats"], rtol=1e-12, atol=0)


def _denoise(i, batch):
    """the stand-in for the sampler the fixture was generated with (make_golden.py: script_denoiser)"""
    return (batch * np.float32(0.5) + np.float32(0.01 * i)).astype(np.float32)


@pytest.mark.parametrize("case", ["small", "ragged"])
def test_tiling_vs_reference_golden(golden, case):
    """split_volume == the reference's own patch extraction / zero padding / (Z,H,W)->(H,W,Z) nest."""
    g = golden("script_tiling.npz")
    vol, ref = g[case + "_vol"], g[case + "_patches_hwz"]
    p, grid = patches.split_volume(vol, 16)
    assert p.shape == (len(ref), 1, 16, 16, 16)
    got = p[:, 0].transpose(0, 2, 3, 1)                 # (Z,H,W) -> (H,W,Z), scripts/test.py:230
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("case", ["small", "ragged"])
def test_stitching_vs_reference_golden(golden, case):
    """stitch_patches == the reference's weighted overlap-add and normalisation, bit for bit wherever
    the total weight is positive; 0 (not uninitialised memory) elsewhere."""
    g = golden("script_tiling.npz")
    vol, ref, wref = g[case + "_vol"], g[case + "_result"], g[case + "_weight"]
    p, grid = patches.split_volume(vol, 16)
    den = [_denoise(i, p[i:i + 1])[0, 0].transpose(1, 2, 0) for i in range(len(p))]   # (H,W,Z), :72
    out, wsum = patches.stitch_patches(den, grid, vol.shape, 16)
    assert np.array_equal(wsum, wref)
    assert np.array_equal(out[wref > 0], ref[wref > 0])
    assert np.all(out[wref == 0] == 0)
    assert (wref == 0).any() and (wref > 0).mean() > 0.7


def test_launcher_geometry_vs_reference_golden(golden):
    """The reference launcher's real shape (110 x 200 x 200 volume, 96^3 patches, 18 of them):
    per-patch checksums and corners of the tiling, a strided sample of the stitched volume."""
    g = golden("script_tiling.npz")
    seed, D, Hh, W = (int(v) for v in g["big_seed_shape"])
    rng = np.random.default_rng(seed)
    vol = (rng.random((D, Hh, W), dtype=np.float32) * 4.0).astype(np.float32)
    p, grid = patches.split_volume(vol, 96)
    hwz = p[:, 0].transpose(0, 2, 3, 1)
    assert len(grid) == 18
    assert np.array_equal(hwz.reshape(18, -1).astype(np.float64).sum(1), g["big_patch_sums"])
    assert np.array_equal(hwz[:, :4, :4, :4], g["big_patch_corner"])
    assert np.array_equal(hwz[:, -3:, -3:, -3:], g["big_patch_last"])
    den = [_denoise(i, p[i:i + 1])[0, 0].transpose(1, 2, 0) for i in range(len(p))]
    out, wsum = patches.stitch_patches(den, grid, vol.shape, 96)
    ws = g["big_weight_strided"]
    assert np.array_equal(wsum[::7, ::7, ::5], ws)
    assert np.array_equal(out[::7, ::7, ::5][ws > 0], g["big_result_strided"][ws > 0])
    assert np.allclose([out.astype(np.float64).sum(), wsum.astype(np.float64).sum()], g["big_result_sum"],
                       rtol=1e-12, atol=0)


def test_start_positions_match_the_reference_constants():
    assert patches.xy_starts(200, 96, 3) == [0, 52, 104]            # scripts/test.py:285-286
    assert patches.xy_starts(150, 96, 3) == [0, 27, 54]             # int(i * (150-96)/2)
    assert patches.xy_starts(96, 96, 1) == [0]
    assert patches.z_starts(90, 96) == [0] and patches.z_starts(96, 96) == [0]
    assert patches.z_starts(130, 96) == [0, 34]
    grid = patches.patch_grid((110, 200, 200), 96)
    assert len(grid) == 18 and grid[0] == (0, 0, 0) and grid[1] == (0, 0, 14) and grid[2] == (0, 52, 0)


def test_split_pads_and_orders_like_the_reference():
    rng = np.random.default_rng(0)
    vol = rng.random((110, 200, 200), dtype=np.float32)              # (D, H, W)
    p, grid = patches.split_volume(vol, 96)
    assert p.shape == (18, 1, 96, 96, 96) and p.dtype == np.float32
    xs, ys, zs = grid[7]
    assert np.array_equal(p[7, 0], vol[zs:zs + 96, xs:xs + 96, ys:ys + 96])
    # a short volume is zero padded along Z
    p2, g2 = patches.split_volume(vol[:90], 96)
    assert len(g2) == 9 and np.all(p2[:, :, 90:] == 0) and np.array_equal(p2[0, 0, :90], vol[:90, :96, :96])
    assert patches.split_volume(vol[None], 96)[0].shape == p.shape    # leading singleton dropped
    with pytest.raises(ValueError):
        patches.split_volume(np.zeros((4, 4)), 96)


def test_hann_window():
    w = patches.hann_window_3d(96)
    assert w.shape == (96, 96, 96) and w.max() == 1.0 and w.min() == 0.0
    h = np.hanning(96)
    assert np.allclose(w[10, 20, 30], h[10] * h[20] * h[30] / (h.max() ** 3))


def test_stitch_is_a_partition_of_unity_inside_the_border():
    """Identity 'denoiser': stitching the input's own patches returns the input wherever the
    total weight is positive; the outermost planes (Hann weight 0) stay 0 as in the reference."""
    rng = np.random.default_rng(1)
    vol = rng.random((110, 200, 200), dtype=np.float32)
    p, grid = patches.split_volume(vol, 96)
    as_hwz = [q[0].transpose(1, 2, 0) for q in p]                     # (Z,H,W) -> (H,W,Z), scripts/test.py:72
    out, wsum = patches.stitch_patches(as_hwz, grid, vol.shape, 96)
    assert out.shape == (200, 200, 110)
    ref = vol.transpose(1, 2, 0)
    inner = wsum > 0
    assert np.allclose(out[inner], ref[inner], atol=1e-5)
    assert np.all(out[~inner] == 0) and not inner[0].any() and not inner[:, :, 0].any()
    assert inner[1:-1, 1:-1, 1:-1].all()                              # only the outer shell has zero weight


def test_volume_io_round_trip(tmp_path):
    vol = np.arange(4 * 5 * 6, dtype=np.float32).reshape(4, 5, 6)
    np.savez(tmp_path / "a.npz", vol)
    np.save(tmp_path / "b.npy", vol[None])
    assert np.array_equal(patches.load_volume(str(tmp_path / "a.npz")), vol)
    assert np.array_equal(patches.load_volume(str(tmp_path / "b.npy")), vol)
    with pytest.raises(ValueError):
        patches.load_volume(str(tmp_path / "c.txt"))


# ---- .tif volumes (guided_diffusion/tiff_io.py: the reference's tifffile.imread / imwrite, scripts/test.py:96, :178, :192).
# tifffile is not importable here and the reference holds no .tif fixture, so the reader and the writer are checked
# against an INDEPENDENT TIFF implementation, Pillow's: files this module writes are decoded by Pillow, files Pillow
# writes (multi-page uint16 / float32, LZW-compressed) are decoded by this module.

def _pil():
    return pytest.importorskip("PIL.Image")


def test_tiff_written_here_is_read_by_pillow_and_back(tmp_path):
    from guided_diffusion import tiff_io
    Image = _pil()
    rng = np.random.default_rng(7)
    vol = rng.normal(size=(5, 12, 9)).astype(np.float32)          # (Z, H, W), H != W
    path = str(tmp_path / "v.tif")
    tiff_io.imwrite(path, vol)
    with Image.open(path) as im:
        assert im.n_frames == 5 and im.size == (9, 12) and im.mode == "F"
        for z in range(5):
            im.seek(z)
            assert np.array_equal(np.array(im), vol[z])
    back = tiff_io.imread(path)
    assert back.dtype == np.float32 and back.shape == vol.shape and np.array_equal(back, vol)
    # big-endian files, integer samples, a single page
    for dt in (np.uint8, np.uint16, np.int16, np.int32, np.float64):
        a = (rng.random((3, 7, 10)) * 100).astype(dt)
        for bo in "<>":
            p = str(tmp_path / ("t_%s_%s.tif" % (np.dtype(dt).name, "le" if bo == "<" else "be")))
            tiff_io.imwrite(p, a, byteorder=bo)
            b = tiff_io.imread(p)
            assert b.dtype == np.dtype(dt) and np.array_equal(a, b), (dt, bo)
    one = str(tmp_path / "one.tif")
    tiff_io.imwrite(one, vol[0])
    assert tiff_io.imread(one).shape == (12, 9)
    with Image.open(str(tmp_path / "t_uint16_be.tif")) as im:       # Pillow agrees on the big-endian file too
        im.seek(2)
        assert np.array_equal(np.array(im), (tiff_io.imread(str(tmp_path / "t_uint16_be.tif")))[2])


def test_tiff_written_by_pillow_is_read_here(tmp_path):
    from guided_diffusion import tiff_io
    Image = _pil()
    rng = np.random.default_rng(8)
    u16 = (rng.random((4, 10, 14)) * 60000).astype(np.uint16)
    f32 = rng.normal(size=(4, 10, 14)).astype(np.float32)
    for name, vol, kw in (("u16.tif", u16, {}), ("f32.tif", f32, {}), ("lzw.tif", u16, {"compression": "tiff_lzw"}),
                          ("zip.tif", f32, {"compression": "tiff_adobe_deflate"})):
        p = str(tmp_path / name)
        frames = [Image.fromarray(v) for v in vol]
        frames[0].save(p, save_all=True, append_images=frames[1:], **kw)
        got = tiff_io.imread(p)
        assert got.shape == vol.shape and got.dtype == vol.dtype and np.array_equal(got, vol), name
    # and through the package's loader: (D, H, W) float32, like load_data_for_worker's vol.astype(np.float32)
    v = patches.load_volume(str(tmp_path / "u16.tif"))
    assert v.dtype == np.float32 and np.array_equal(v, u16.astype(np.float32))


def test_tiff_reader_refuses_what_it_cannot_decode(tmp_path):
    from guided_diffusion import tiff_io
    bad = tmp_path / "bad.tif"
    bad.write_bytes(b"not a tiff at all")
    with pytest.raises(tiff_io.TiffError, match="not a TIFF"):
        tiff_io.imread(str(bad))
    good = str(tmp_path / "g.tif")
    tiff_io.imwrite(good, np.zeros((2, 4, 4), np.float32))
    raw = bytearray(open(good, "rb").read())
    cut = tmp_path / "cut.tif"
    cut.write_bytes(bytes(raw[:len(raw) - 40]))                     # the last page's strip leaves the file
    with pytest.raises(tiff_io.TiffError, match="leave"):
        tiff_io.imread(str(cut))
    with pytest.raises(tiff_io.TiffError, match="unsupported dtype"):
        tiff_io.imwrite(good, np.zeros((2, 4, 4), np.complex64))
    # an ImageJ hyperstack: ONE directory, planes contiguous behind it, "images=N" in the description
    import struct
    planes = np.arange(3 * 4 * 5, dtype=">u2").reshape(3, 4, 5)
    desc = b"ImageJ=1.53\nimages=3\nslices=3\n\0"
    ents = [(256, 3, 1, 5), (257, 3, 1, 4), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (270, 2, len(desc), None),
            (273, 4, 1, None), (277, 3, 1, 1), (278, 3, 1, 4), (279, 4, 1, 4 * 5 * 2)]
    ifd_len = 2 + 12 * len(ents) + 4
    desc_off = 8 + ifd_len
    data_off = desc_off + len(desc) + (len(desc) & 1)
    out = struct.pack(">2sHI", b"MM", 42, 8) + struct.pack(">H", len(ents))
    for tag, typ, cnt, val in ents:
        if tag == 270:
            field = struct.pack(">I", desc_off)
        elif tag == 273:
            field = struct.pack(">I", data_off)
        else:
            field = struct.pack(">H" if typ == 3 else ">I", val).ljust(4, b"\0")
        out += struct.pack(">HHI", tag, typ, cnt) + field
    out += struct.pack(">I", 0) + desc + (b"\0" if len(desc) & 1 else b"") + planes.tobytes()
    ij = tmp_path / "ij.tif"
    ij.write_bytes(out)
    got = tiff_io.imread(str(ij))
    assert got.shape == (3, 4, 5) and np.array_equal(got, planes.astype(np.uint16))
