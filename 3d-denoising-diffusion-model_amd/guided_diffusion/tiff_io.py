"""
TIFF volumes for the inference entry: what scripts/test.py asks of `tifffile` -- `imread(path)` of a
(Z, H, W) stack (scripts/test.py:96, :192) and `imwrite(path, float32 (Z, H, W))` (:178) -- without the
package (it is not importable in this image, and the script's .tif branch was dead code here until r04).

Read:  classic TIFF and BigTIFF, either byte order, one image per IFD (what tifffile and ImageJ write for a
       stack) or ImageJ's single-IFD hyperstack ("images=N" in the description, planes contiguous), one
       sample per pixel, 8/16/32/64-bit unsigned / signed / IEEE samples, uncompressed strips.  Pages that are
       compressed or tiled go through Pillow when it is importable; otherwise they are refused by name.
Write: little-endian, one uncompressed strip per page, float32 (or any of the sample types above), the
       `{"shape": [...]}` description tifffile itself writes on the first page, BigTIFF above 4 GiB.

When `tifffile` IS importable it is used, as in the reference.
"""

import json
import struct

import numpy as np

_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "c", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
          13: "I", 16: "Q", 17: "q", 18: "Q"}
_SAMPLE = {(1, 8): "u1", (1, 16): "u2", (1, 32): "u4", (1, 64): "u8", (2, 8): "i1", (2, 16): "i2", (2, 32): "i4",
           (2, 64): "i8", (3, 16): "f2", (3, 32): "f4", (3, 64): "f8"}

IMAGE_WIDTH, IMAGE_LENGTH, BITS, COMPRESSION, PHOTOMETRIC, DESCRIPTION, STRIP_OFFSETS = 256, 257, 258, 259, 262, 270, 273
SAMPLES_PER_PIXEL, ROWS_PER_STRIP, STRIP_COUNTS, XRES, YRES, PLANAR, RES_UNIT, SOFTWARE = 277, 278, 279, 282, 283, 284, 296, 305
TILE_WIDTH, SAMPLE_FORMAT = 322, 339


class TiffError(ValueError):
    pass


def _ifds(buf):
    """[(tags: {tag: tuple of values})] of every IFD in file order, and the byte-order character"""
    if len(buf) < 8 or buf[:2] not in (b"II", b"MM"):
        raise TiffError("not a TIFF file (no II / MM byte-order mark)")
    bo = "<" if buf[:2] == b"II" else ">"
    magic = struct.unpack_from(bo + "H", buf, 2)[0]
    if magic == 42:
        big, off = False, struct.unpack_from(bo + "I", buf, 4)[0]
    elif magic == 43:
        big, off = True, struct.unpack_from(bo + "Q", buf, 8)[0]
    else:
        raise TiffError("not a TIFF file (magic %d)" % magic)
    cnt_f, ent, val_f, nxt_f = ("Q", 20, 8, "Q") if big else ("H", 12, 4, "I")
    out, seen = [], set()
    while off:
        if off in seen or off + struct.calcsize(cnt_f) > len(buf):
            raise TiffError("corrupt TIFF: IFD chain leaves the file or loops")
        seen.add(off)
        n = struct.unpack_from(bo + cnt_f, buf, off)[0]
        p = off + struct.calcsize(cnt_f)
        tags = {}
        for _ in range(n):
            tag, typ = struct.unpack_from(bo + "HH", buf, p)
            count = struct.unpack_from(bo + ("Q" if big else "I"), buf, p + 4)[0]
            code = _TYPES.get(typ)
            if code is not None:
                size = struct.calcsize("=" + code) * count
                vp = p + (12 if big else 8)
                if size > val_f:
                    vp = struct.unpack_from(bo + ("Q" if big else "I"), buf, vp)[0]
                if vp + size > len(buf):
                    raise TiffError("corrupt TIFF: tag %d points outside the file" % tag)
                if code == "c":
                    tags[tag] = (bytes(buf[vp:vp + size]),)
                else:
                    tags[tag] = struct.unpack_from(bo + code * count, buf, vp)
            p += ent
        out.append(tags)
        off = struct.unpack_from(bo + nxt_f, buf, p)[0]
    if not out:
        raise TiffError("TIFF file without an image directory")
    return out, bo


def _page_dtype(tags, bo):
    spp = tags.get(SAMPLES_PER_PIXEL, (1,))[0]
    if spp != 1:
        raise TiffError("only one sample per pixel is supported (this page has %d)" % spp)
    bits = tags.get(BITS, (1,))[0]
    fmt = tags.get(SAMPLE_FORMAT, (1,))[0]
    code = _SAMPLE.get((fmt if fmt in (1, 2, 3) else 1, bits))
    if code is None:
        raise TiffError("unsupported sample type: %d bits, SampleFormat %d" % (bits, fmt))
    return np.dtype(bo + code)


def _plain(tags):
    return tags.get(COMPRESSION, (1,))[0] == 1 and TILE_WIDTH not in tags and STRIP_OFFSETS in tags


def _read_plain_page(buf, tags, bo, planes=1):
    w, h = tags[IMAGE_WIDTH][0], tags[IMAGE_LENGTH][0]
    dt = _page_dtype(tags, bo)
    offs, cnts = tags[STRIP_OFFSETS], tags.get(STRIP_COUNTS)
    need = planes * h * w * dt.itemsize
    if planes > 1 or len(offs) == 1:
        # one strip (or ImageJ's contiguous hyperstack, whose byte count only covers the first plane)
        if offs[0] + need > len(buf):
            raise TiffError("corrupt TIFF: image data leave the file")
        a = np.frombuffer(buf, dtype=dt, count=planes * h * w, offset=offs[0])
        return a.reshape((planes, h, w) if planes > 1 else (h, w))
    if cnts is None or len(cnts) != len(offs):
        raise TiffError("corrupt TIFF: StripByteCounts does not match StripOffsets")
    parts = []
    for o, c in zip(offs, cnts):
        if o + c > len(buf):
            raise TiffError("corrupt TIFF: a strip leaves the file")
        parts.append(buf[o:o + c])
    raw = b"".join(parts)
    if len(raw) < need:
        raise TiffError("corrupt TIFF: strips hold %d bytes, the page needs %d" % (len(raw), need))
    return np.frombuffer(raw, dtype=dt, count=h * w).reshape(h, w)


def _imagej_planes(tags):
    d = tags.get(DESCRIPTION)
    if not d or not d[0].startswith(b"ImageJ"):
        return 1
    for line in d[0].split(b"\n"):
        if line.startswith(b"images="):
            try:
                return max(1, int(line[7:].strip(b"\0 \r")))
            except ValueError:
                return 1
    return 1


def _read_with_pillow(path, why):
    try:
        from PIL import Image
    except ImportError as e:
        raise TiffError("%s: %s, and Pillow is not importable to decode it" % (path, why)) from e
    pages = []
    with Image.open(path) as im:
        for i in range(getattr(im, "n_frames", 1)):
            im.seek(i)
            pages.append(np.array(im))
    return pages


def imread(path):
    """The file's image stack as tifffile.imread returns it for such files: (pages, H, W), or (H, W) for a single page;
    native byte order."""
    try:
        import tifffile
        return tifffile.imread(path)
    except ImportError:
        pass
    with open(path, "rb") as f:
        buf = f.read()
    ifds, bo = _ifds(buf)
    if all(_plain(t) for t in ifds):
        nj = _imagej_planes(ifds[0]) if len(ifds) == 1 else 1
        pages = [_read_plain_page(buf, t, bo, nj if i == 0 else 1) for i, t in enumerate(ifds)]
        if nj > 1:
            pages = list(pages[0])
    else:
        comp = sorted({t.get(COMPRESSION, (1,))[0] for t in ifds if not _plain(t)})
        pages = _read_with_pillow(path, "compressed or tiled pages (Compression %s)" % comp)
    shapes = {p.shape for p in pages}
    if len(shapes) != 1:
        raise TiffError("%s: pages of different shapes %s" % (path, sorted(shapes)))
    vol = np.stack(pages) if len(pages) > 1 else pages[0]
    return np.ascontiguousarray(vol.astype(vol.dtype.newbyteorder("="), copy=False))


def imwrite(path, data, byteorder="<"):
    """(pages, H, W) or (H, W) array -> multi-page TIFF, one uncompressed strip per page (scripts/test.py:178 writes the
    float32 (Z, H, W) volume this way).  `byteorder` ">" is there for the reader's tests."""
    try:
        import tifffile
        if byteorder == "<":
            return tifffile.imwrite(path, data)
    except ImportError:
        pass
    a = np.asarray(data)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise TiffError("expected (pages, H, W) or (H, W), got shape %s" % (a.shape,))
    kind = {"u": 1, "i": 2, "f": 3}.get(a.dtype.kind)
    if kind is None or (kind, a.dtype.itemsize * 8) not in _SAMPLE:
        raise TiffError("unsupported dtype %s" % a.dtype)
    bo = byteorder
    a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder(bo), copy=False))
    n, h, w = a.shape
    plane = h * w * a.dtype.itemsize
    big = n * (plane + 512) + 4096 >= 1 << 32
    desc = json.dumps({"shape": list(np.asarray(data).shape)}).encode() + b"\0"
    soft = b"ddpm3d tiff_io\0"
    word = "Q" if big else "I"

    def entry(tag, typ, count, value_bytes=None, value=None):
        # value: a scalar that fits the entry; value_bytes: data placed behind the IFD (offset patched by the caller)
        return (tag, typ, count, value, value_bytes)

    with open(path, "wb") as f:
        if big:
            f.write(struct.pack(bo + "2sHHHQ", b"II" if bo == "<" else b"MM", 43, 8, 0, 16))
        else:
            f.write(struct.pack(bo + "2sHI", b"II" if bo == "<" else b"MM", 42, 8))
        pos = f.tell()
        for i in range(n):
            ents = [entry(IMAGE_WIDTH, 4, 1, value=w), entry(IMAGE_LENGTH, 4, 1, value=h),
                    entry(BITS, 3, 1, value=a.dtype.itemsize * 8), entry(COMPRESSION, 3, 1, value=1),
                    entry(PHOTOMETRIC, 3, 1, value=1)]
            if i == 0:
                ents.append(entry(DESCRIPTION, 2, len(desc), value_bytes=desc))
            ents += [entry(STRIP_OFFSETS, 16 if big else 4, 1, value="data"), entry(SAMPLES_PER_PIXEL, 3, 1, value=1),
                     entry(ROWS_PER_STRIP, 4, 1, value=h), entry(STRIP_COUNTS, 16 if big else 4, 1, value=plane),
                     entry(XRES, 5, 1, value_bytes=struct.pack(bo + "II", 1, 1)),
                     entry(YRES, 5, 1, value_bytes=struct.pack(bo + "II", 1, 1)),
                     entry(PLANAR, 3, 1, value=1), entry(RES_UNIT, 3, 1, value=1),
                     entry(SOFTWARE, 2, len(soft), value_bytes=soft), entry(SAMPLE_FORMAT, 3, 1, value=kind)]
            ifd_bytes = (8 + 20 * len(ents) + 8) if big else (2 + 12 * len(ents) + 4)
            extra_pos = pos + ifd_bytes
            extras = b""
            recs = []
            for tag, typ, count, value, vb in ents:
                if vb is not None and len(vb) <= (8 if big else 4):
                    field = vb.ljust(8 if big else 4, b"\0")
                elif vb is not None:
                    field = struct.pack(bo + word, extra_pos + len(extras))
                    extras += vb + (b"\0" if len(vb) & 1 else b"")
                else:
                    field = None
                recs.append((tag, typ, count, value, field))
            data_pos = extra_pos + len(extras)
            data_pos += -data_pos % 16
            next_ifd = data_pos + plane
            next_ifd += -next_ifd % 16
            out = struct.pack(bo + ("Q" if big else "H"), len(recs))
            for tag, typ, count, value, field in recs:
                if field is None:
                    v = data_pos if value == "data" else value
                    code = {3: "H", 4: "I", 16: "Q"}[typ]
                    field = struct.pack(bo + code, v).ljust(8 if big else 4, b"\0")
                out += struct.pack(bo + "HH" + word, tag, typ, count) + field
            out += struct.pack(bo + word, next_ifd if i + 1 < n else 0)
            assert len(out) == ifd_bytes
            f.write(out + extras)
            f.write(b"\0" * (data_pos - f.tell()))
            f.write(a[i].tobytes())
            if i + 1 < n:
                f.write(b"\0" * (next_ifd - f.tell()))
            pos = next_ifd
