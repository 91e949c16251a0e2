"""CPU suite: JPEG-in-TIFF (Compression 7) in unet_amd/tiffio.py -- csrc/host/tiff_jpeg.cpp behind include/unet_tiff.h.

The reference reads rasters through GDAL (create_tiles_unet.py:252-434, data.py:18-28), i.e. libtiff + libjpeg; the checker here is that very
pair through Pillow's bindings (libtiff for TIFF files Pillow wrote, libjpeg for JPEG streams wrapped into TIFFs by hand), and the bar is
0 differing bytes: the decoder restates libjpeg's integer IDCT, fancy upsampling and colour tables rather than any float formulation."""
import ctypes as C
import io
import re
import struct
import time
from pathlib import Path

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image, features  # noqa: E402

from unet_amd.tiffio import read_tiff  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
needs_libtiff = pytest.mark.skipif(not features.check("libtiff"), reason="Pillow without libtiff")


def _scene(h, w, c, seed=0, noise=12.0):
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([np.sin(x / (7.0 + k)) * 60 + np.cos(y / (5.0 + 2 * k)) * 50 + 128 for k in range(c)], -1)
    return np.clip(a + g.normal(0, noise, a.shape), 0, 255).astype(np.uint8)


def _wrap(path, blocks, W, H, spp, photometric, *, tile=None, rows_per_strip=None, planar=1, tables=None, extra=()):
    """a classic little-endian TIFF around ready-made JPEG streams (one per strip / tile, in file order)"""
    entries = []          # (tag, type, count, bytes)

    def add(tag, typ, vals):
        fmt = {1: "B", 3: "H", 4: "I", 7: "B"}[typ]
        entries.append((tag, typ, len(vals), struct.pack("<" + fmt * len(vals), *vals)))
    add(256, 4, [W]); add(257, 4, [H]); add(258, 3, [8] * spp); add(259, 3, [7]); add(262, 3, [photometric]); add(277, 3, [spp]); add(284, 3, [planar])
    if spp > 3 or (spp == 2):
        add(338, 3, [0] * (spp - (3 if spp > 3 else 1)))
    if tables is not None:
        add(347, 7, list(tables))
    for t in extra:
        add(*t)
    n = len(blocks)
    if tile:
        add(322, 4, [tile[0]]); add(323, 4, [tile[1]])
        off_tag, cnt_tag = 324, 325
    else:
        add(278, 4, [rows_per_strip or H])
        off_tag, cnt_tag = 273, 279
    add(off_tag, 4, [0] * n); add(cnt_tag, 4, [len(b) for b in blocks])
    entries.sort(key=lambda e: e[0])
    ifd_size = 2 + 12 * len(entries) + 4
    data_off = 8 + ifd_size
    big = b""
    recs = []
    for tag, typ, cnt, raw in entries:
        if len(raw) <= 4:
            recs.append([tag, typ, cnt, raw.ljust(4, b"\0"), None])
        else:
            recs.append([tag, typ, cnt, None, len(big)])
            big += raw + (b"\0" if len(raw) & 1 else b"")
    pix_off = data_off + len(big)
    offs, o = [], pix_off
    for b in blocks:
        offs.append(o)
        o += len(b) + (len(b) & 1)
    raw_offs = struct.pack("<" + "I" * n, *offs)
    out = bytearray(b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", len(entries)))
    big = bytearray(big)
    for tag, typ, cnt, inline, rel in recs:
        if tag == off_tag:
            if rel is None:
                inline = raw_offs.ljust(4, b"\0")
            else:
                big[rel:rel + len(raw_offs)] = raw_offs
        out += struct.pack("<HHI", tag, typ, cnt) + (inline if rel is None else struct.pack("<I", data_off + rel))
    out += struct.pack("<I", 0) + big
    for b in blocks:
        out += b + (b"\0" if len(b) & 1 else b"")
    Path(path).write_bytes(bytes(out))


def _jpeg(arr, mode, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr if arr.ndim == 2 or arr.shape[-1] > 1 else arr[..., 0], mode).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _libjpeg(stream, mode=None):
    im = Image.open(io.BytesIO(stream))
    if mode:
        im.draft(mode, im.size)
    a = np.asarray(im)
    return a if a.ndim == 3 else a[..., None]


def _chw(a):          # read_tiff returns [C, H, W] (or [H, W])
    return a[None] if a.ndim == 2 else a


def test_host_library_exports_what_unet_tiff_h_declares():
    from unet_amd.build import build_host_codecs
    lib = C.CDLL(str(build_host_codecs()))
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "unet_tiff.h").read_text(), flags=re.S)
    names = re.findall(r"\b(unet_\w+)\s*\(", text)
    assert names == ["unet_tiff_jpeg_decode"]
    for n in names + ["unet_tiff_lzw_decode", "unet_tiff_packbits_decode"]:
        assert hasattr(lib, n), n
    # argument validation without any data
    fn = lib.unet_tiff_jpeg_decode
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    assert fn(None, 0, None, 0, 0, None, 0, None) == -1


@needs_libtiff
@pytest.mark.parametrize("mode,shape,kw", [
    ("L", (64, 64, 1), {}),
    ("L", (100, 77, 1), {"quality": 90}),
    ("RGB", (101, 75, 3), {"quality": 95}),
    ("RGB", (300, 210, 3), {"quality": 50}),          # three strips, the last one short
    ("RGBA", (130, 131, 4), {}),                      # 4-band imagery: the reference's tiles (params_and_main.py: 4 input channels)
    ("CMYK", (257, 66, 4), {"quality": 30}),
])
def test_files_written_by_libtiff_read_back_like_libtiff_reads_them(tmp_path, mode, shape, kw):
    arr = _scene(*shape, seed=shape[0])
    p = tmp_path / "a.tif"
    Image.fromarray(arr if shape[2] > 1 else arr[..., 0], mode).save(p, format="TIFF", compression="jpeg", **kw)
    ref = np.asarray(Image.open(p))
    ref = ref if ref.ndim == 3 else ref[..., None]
    got, _ = read_tiff(p)
    assert np.array_equal(_chw(got), np.moveaxis(ref, -1, 0))
    assert np.abs(ref.astype(int) - arr.astype(int)).mean() < 12          # (and it is the picture, not noise)


@needs_libtiff
def test_random_content_and_sizes_against_libtiff(tmp_path):
    g = np.random.default_rng(5)
    for k in range(12):
        h, w = int(g.integers(1, 150)), int(g.integers(1, 150))
        c = int(g.choice([1, 3, 4]))
        arr = g.integers(0, 256, (h, w, c), dtype=np.uint8) if k % 2 else _scene(h, w, c, seed=k, noise=40.0)
        p = tmp_path / f"r{k}.tif"
        Image.fromarray(arr if c > 1 else arr[..., 0], {1: "L", 3: "RGB", 4: "RGBA"}[c]).save(p, format="TIFF", compression="jpeg",
                                                                                                quality=int(g.integers(5, 100)))
        ref = np.asarray(Image.open(p))
        ref = ref if ref.ndim == 3 else ref[..., None]
        got, _ = read_tiff(p)
        assert np.array_equal(_chw(got), np.moveaxis(ref, -1, 0)), (k, h, w, c)


@needs_libtiff
def test_ycbcr_photometric_written_by_libtiff_comes_back_as_rgb(tmp_path):
    """PhotometricInterpretation 6 without subsampling: libtiff (asked for JPEGCOLORMODE_RGB, as GDAL and Pillow ask) converts, so does read_tiff"""
    rgb = _scene(90, 123, 3, seed=3)
    p = tmp_path / "y.tif"
    Image.fromarray(rgb, "RGB").convert("YCbCr").save(p, format="TIFF", compression="jpeg", quality=85)
    im = Image.open(p)
    ref = np.asarray(im.convert("RGB")) if im.mode != "RGB" else np.asarray(im)
    from unet_amd.tiffio import _parse_tags
    tags, _ = _parse_tags(p.read_bytes(), p)
    assert tags[262][0] == 6
    got, _ = read_tiff(p)
    if im.mode == "RGB":          # libtiff did the conversion: the bytes are libjpeg's
        assert np.array_equal(got, np.moveaxis(ref, -1, 0))
    else:                          # Pillow kept raw YCbCr and converted itself (other rounding): within 2 counts
        assert np.abs(got.astype(int) - np.moveaxis(ref, -1, 0).astype(int)).max() <= 2
    assert np.abs(got.astype(int) - np.moveaxis(rgb, -1, 0).astype(int)).mean() < 10


@pytest.mark.parametrize("sub,size", [(0, (50, 70)), (1, (64, 64)), (1, (37, 51)), (2, (64, 64)), (2, (75, 101)), (2, (33, 18)),
                                      (1, (9, 4)), (2, (9, 3)), (2, (1, 1)), (2, (17, 2))])
def test_subsampled_chroma_streams_equal_libjpeg(tmp_path, sub, size):
    """4:4:4 / 4:2:2 / 4:2:0 streams written by libjpeg, wrapped as one-strip YCbCr TIFFs: fancy upsampling (and plain replication for
    components <= 2 samples wide) + colour conversion equal libjpeg's own decode of the stream"""
    h, w = size
    rgb = _scene(h, w, 3, seed=h * w)
    s = _jpeg(rgb, "RGB", quality=88, subsampling=sub)
    ref = _libjpeg(s)
    p = tmp_path / "s.tif"
    _wrap(p, [s], w, h, 3, 6, extra=[(530, 3, [(1, 2, 2)[sub], (1, 1, 2)[sub]])])
    got, _ = read_tiff(p)
    assert np.array_equal(got, np.moveaxis(ref, -1, 0))
    if features.check("libtiff") and w > 2:          # and libtiff reads the hand-made file the same way
        via = Image.open(p)
        if via.mode == "RGB":
            assert np.array_equal(np.asarray(via), ref)


def test_tiles_planes_restart_intervals_and_shared_tables(tmp_path):
    g = np.random.default_rng(2)
    # tiled, 3 x 2 tiles of 64 x 48 over a 150 x 100 scene, 4 bands without colour transform, restart markers every 3 MCUs
    H, W, th, tw = 100, 150, 48, 64
    arr = _scene(H, W, 4, seed=9)
    blocks, ref = [], np.zeros((H, W, 4), np.uint8)
    for j in range(-(-H // th)):
        for i in range(-(-W // tw)):
            t = np.zeros((th, tw, 4), np.uint8)
            part = arr[j * th:(j + 1) * th, i * tw:(i + 1) * tw]
            t[:part.shape[0], :part.shape[1]] = part
            s = _jpeg(t, "CMYK", quality=int(g.integers(40, 95)), restart_marker_blocks=3)
            assert s.count(b"\xff\xdd") == 1
            d = 255 - _libjpeg(s)          # Pillow's JPEG plugin stores CMYK inverted (Adobe) and inverts again on load: the stream's samples are these
            ref[j * th:(j + 1) * th, i * tw:(i + 1) * tw] = d[:part.shape[0], :part.shape[1]]
            blocks.append(s)
    _wrap(tmp_path / "t.tif", blocks, W, H, 4, 5, tile=(tw, th))
    got, _ = read_tiff(tmp_path / "t.tif")
    assert np.array_equal(got, np.moveaxis(ref, -1, 0))

    # PlanarConfiguration 2: one single-component stream per band and strip (the per-component block walk), widths off the 8-grid
    H, W, rps = 45, 61, 16
    arr = _scene(H, W, 3, seed=4)
    blocks, ref = [], np.zeros((3, H, W), np.uint8)
    for c in range(3):
        for r0 in range(0, H, rps):
            s = _jpeg(arr[r0:r0 + rps, :, c], "L", quality=80, restart_marker_rows=1)
            ref[c, r0:r0 + rps] = _libjpeg(s)[..., 0]
            blocks.append(s)
    _wrap(tmp_path / "p.tif", blocks, W, H, 3, 2, rows_per_strip=rps, planar=2)
    got, _ = read_tiff(tmp_path / "p.tif")
    assert np.array_equal(got, ref)

    # abbreviated streams: the quantisation / Huffman tables moved out of the strips into the JPEGTables tag (what libtiff and GDAL write)
    H, W, rps = 70, 90, 32
    arr = _scene(H, W, 3, seed=6)
    blocks, ref, tables = [], np.zeros((H, W, 3), np.uint8), None
    for r0 in range(0, H, rps):
        s = _jpeg(arr[r0:r0 + rps], "RGB", quality=75, subsampling=0)
        ref[r0:r0 + rps] = _libjpeg(s)
        segs, body, q = [], b"", 2
        while s[q + 1] != 0xDA:
            ln = struct.unpack(">H", s[q + 2:q + 4])[0]
            if s[q + 1] in (0xDB, 0xC4):
                segs.append(s[q:q + 2 + ln])
            else:
                body += s[q:q + 2 + ln]
            q += 2 + ln
        t = b"\xff\xd8" + b"".join(segs) + b"\xff\xd9"
        assert tables in (None, t)          # same quality: same tables in every strip
        tables = t
        blocks.append(b"\xff\xd8" + body + s[q:])
    _wrap(tmp_path / "j.tif", blocks, W, H, 3, 6, rows_per_strip=rps, tables=tables, extra=[(530, 3, [1, 1])])
    got, _ = read_tiff(tmp_path / "j.tif")
    assert np.array_equal(got, np.moveaxis(ref, -1, 0))
    with pytest.raises(ValueError):          # ... and without the tag the strips cannot be decoded
        _wrap(tmp_path / "k.tif", blocks, W, H, 3, 6, rows_per_strip=rps, extra=[(530, 3, [1, 1])])
        read_tiff(tmp_path / "k.tif")


def test_refusals_are_loud(tmp_path):
    rgb = _scene(40, 40, 3, seed=1)
    s = _jpeg(rgb, "RGB", quality=80)
    _wrap(tmp_path / "cut.tif", [s[:len(s) // 2]], 40, 40, 3, 6)
    with pytest.raises(ValueError):
        read_tiff(tmp_path / "cut.tif")
    _wrap(tmp_path / "garbage.tif", [bytes(200)], 40, 40, 3, 6)
    with pytest.raises(ValueError):
        read_tiff(tmp_path / "garbage.tif")
    _wrap(tmp_path / "prog.tif", [_jpeg(rgb, "RGB", progressive=True)], 40, 40, 3, 6)
    with pytest.raises(NotImplementedError):
        read_tiff(tmp_path / "prog.tif")
    _wrap(tmp_path / "size.tif", [s], 40, 48, 3, 6)          # the frame holds fewer rows than the strip must
    with pytest.raises(ValueError):
        read_tiff(tmp_path / "size.tif")
    _wrap(tmp_path / "bands.tif", [s], 40, 40, 4, 2)         # 3 components in a 4-band file
    with pytest.raises(ValueError):
        read_tiff(tmp_path / "bands.tif")
    k = s.index(b"\xff\xc0")                                 # a damaged frame header: 65535 x 65535 -- refused from the header alone
    big = s[:k + 5] + b"\xff\xff\xff\xff" + s[k + 9:]         # (found by tests/fuzz/tiff_fuzz.cpp: it used to allocate 4 GB planes and decode for minutes)
    _wrap(tmp_path / "huge.tif", [big], 40, 40, 3, 6)
    t0 = time.perf_counter()
    with pytest.raises(ValueError):
        read_tiff(tmp_path / "huge.tif")
    assert time.perf_counter() - t0 < 1.0
