"""Minimal GeoTIFF reader / writer (no GDAL, no rasterio, no tifffile: none of them is installed here).

Covers what the tile workflow of the reference needs (``data.py:18-28`` reads tiles with rasterio, ``predict.py:19-52``
writes them with GDAL; ``create_tiles_unet.py:252-434`` opens whole scenes with GDAL): classic TIFF and BigTIFF, little/big endian,
strips or tiles, chunky or planar, 8/16/32/64-bit unsigned / signed / float samples, the compressions GDAL writes by default or
on request -- none, LZW (5), Deflate (8 / 32946), PackBits (32773) -- with horizontal differencing (Predictor 2), plus the GeoTIFF
georeferencing tags (ModelPixelScale 33550, ModelTiepoint 33922, GeoKeyDirectory 34735, GeoDoubleParams 34736, GeoAsciiParams
34737, GDAL_NODATA 42113) which are passed through verbatim.  Floating-point differencing (Predictor 3) is undone on float samples.
JPEG-in-TIFF (Compression 7 with the JPEGTables tag: baseline 8-bit, what GDAL's COMPRESS=JPEG writes; PhotometricInterpretation 6 comes back
as RGB, as it does from GDAL) is decoded to the bytes libtiff + libjpeg produce (csrc/host/tiff_jpeg.cpp).  Anything else (old-style JPEG 6,
12-bit / progressive JPEG, CCITT, ...) raises loudly.
The byte-oriented decoders are C functions (csrc/tiff_codecs.hip: in libunet_hip.so, and linked by g++ alone -- together with the JPEG
decoder -- into the host-only libunet_tiff.so); files are memory-mapped for reading and written uncompressed.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4),
          10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}
GEO_TAGS = (33550, 33922, 34735, 34736, 34737, 42113)


def _dtype(bits: int, fmt: int, bo: str) -> np.dtype:
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt, "u")
    return np.dtype(f"{bo}{kind}{bits // 8}")


def _parse_tags(b, path):
    """IFD 0 of a TIFF (classic: magic 42, 32-bit offsets; BigTIFF: magic 43, 64-bit offsets and counts) held in a bytes-like object
    (bytes or mmap): {tag: values}, byte-order prefix"""
    bo = {b"II": "<", b"MM": ">"}[bytes(b[:2])]
    magic = struct.unpack(bo + "H", b[2:4])[0]
    if magic == 42:
        off = struct.unpack(bo + "I", b[4:8])[0]
        n = struct.unpack(bo + "H", b[off:off + 2])[0]
        base, esz, cfmt, inl = off + 2, 12, "I", 4
    elif magic == 43:
        osz, zero = struct.unpack(bo + "HH", b[4:8])
        if osz != 8 or zero != 0:
            raise NotImplementedError(f"{path}: BigTIFF with offset size {osz}")
        off = struct.unpack(bo + "Q", b[8:16])[0]
        n = struct.unpack(bo + "Q", b[off:off + 8])[0]
        base, esz, cfmt, inl = off + 8, 20, "Q", 8
    else:
        raise NotImplementedError(f"{path}: not a TIFF (magic {magic})")
    tags: Dict[int, tuple] = {}
    for i in range(n):
        e = bytes(b[base + esz * i: base + esz * (i + 1)])
        tag, typ = struct.unpack(bo + "HH", e[:4])
        cnt = struct.unpack(bo + cfmt, e[4:4 + inl])[0]
        if typ not in _TYPES:
            continue
        fmt, sz = _TYPES[typ]
        total = sz * cnt
        if total <= inl:
            data = e[4 + inl:4 + inl + total]
        else:
            o = struct.unpack(bo + cfmt, e[4 + inl:4 + 2 * inl])[0]
            if o + total > len(b):          # (a damaged count would otherwise become a format string of that many characters)
                raise ValueError(f"{path}: tag {tag} holds {cnt} values at offset {o}: past the end of the file ({len(b)} bytes)")
            data = bytes(b[o:o + total])
        if typ == 2:
            tags[tag] = (data.rstrip(b"\0").decode("latin1"),)
        else:
            tags[tag] = struct.unpack(bo + fmt[0] * (cnt * len(fmt)), data)
    return tags, bo


_codec_lib = None


def _codecs():
    """the two byte-oriented TIFF decoders (csrc/tiff_codecs.hip: plain C++, no device code).  ``python -m unet_amd.build`` also links them
    into a host-only ``libunet_tiff.so`` (g++), so a tile-preparation box without the HIP runtime reads the rasters GDAL writes by default;
    libunet_hip.so exports the same two symbols."""
    global _codec_lib
    if _codec_lib is None:
        import ctypes as C
        host = Path(__file__).resolve().parent / "lib" / "libunet_tiff.so"
        if host.exists():
            lib = C.CDLL(str(host))
        else:
            from ._lib import lib
        for fn in (lib.unet_tiff_lzw_decode, lib.unet_tiff_packbits_decode):
            fn.restype, fn.argtypes = C.c_longlong, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
        try:                       # include/unet_tiff.h: the JPEG decoder lives in the host-only library alone
            lib.unet_tiff_jpeg_decode.restype = C.c_longlong
            lib.unet_tiff_jpeg_decode.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
            lib.has_jpeg = True
        except AttributeError:     # libunet_hip.so, or a libunet_tiff.so built before the decoder existed
            lib.has_jpeg = False
        _codec_lib = lib
    return _codec_lib


def _decoder(comp: int, path):
    """(uint8 array view of one compressed strip / tile, capacity) -> decoded bytes (a uint8 array of at most `cap` bytes)"""
    if comp == 1:
        return lambda raw, cap: raw
    if comp in (8, 32946):
        import zlib

        def inflate(raw, cap):
            d = zlib.decompressobj()
            out = d.decompress(raw, cap)                   # bounded: a crafted strip cannot expand beyond what the image needs
            if d.unconsumed_tail and d.decompress(d.unconsumed_tail, 1):
                raise ValueError(f"{path}: Deflate strip decodes to more than the {cap} bytes its rows hold")
            return np.frombuffer(out, dtype=np.uint8)
        return inflate
    if comp in (5, 32773):
        lib = _codecs()
        fn = lib.unet_tiff_lzw_decode if comp == 5 else lib.unet_tiff_packbits_decode

        def dec(raw, cap):
            src = np.ascontiguousarray(raw)
            dst = np.empty(cap, dtype=np.uint8)
            got = fn(src.ctypes.data, src.size, dst.ctypes.data, cap)       # (ctypes releases the GIL: the feed's decode pool runs these in parallel)
            if got < 0:
                raise ValueError(f"{path}: corrupt {'LZW' if comp == 5 else 'PackBits'} data")
            return dst[:got]
        return dec
    raise NotImplementedError(f"{path}: TIFF compression {comp} is not supported (none, LZW, Deflate, PackBits and new-style JPEG (7) are)")


def _jpeg_decoder(tags, path):
    """(compressed strip / tile, rows, cols, samples per pixel) -> uint8 [rows', cols', samples] with the frame size the JPEG stream itself states
    (libtiff writes a short last strip as a short frame, tiles always whole).  TIFF Technical Note 2: tag 347 holds the shared tables."""
    import ctypes as C
    lib = _codecs()
    if not lib.has_jpeg:
        raise RuntimeError(f"{path}: JPEG-in-TIFF needs unet_amd/lib/libunet_tiff.so (python -m unet_amd.build)")
    tables = np.array(tags[347], dtype=np.uint8) if 347 in tags else None
    ycc = 1 if tags.get(262, (1,))[0] == 6 else 0

    import threading
    tl = threading.local()          # one output buffer per decoding thread (a fresh >= 128 KB array per strip is an mmap + its page faults, and
                                    # those serialise the threads of a large scene); the caller copies the block out before its next call

    def dec(raw, rows, cols, pix):
        src = np.ascontiguousarray(raw)
        cap = rows * cols * pix
        dst = getattr(tl, "buf", None)
        if dst is None or dst.size < cap:
            dst = tl.buf = np.empty(cap, dtype=np.uint8)
        dims = (C.c_int * 3)()
        got = lib.unet_tiff_jpeg_decode(None if tables is None else tables.ctypes.data, 0 if tables is None else tables.size, src.ctypes.data,
                                        src.size, ycc, dst.ctypes.data, cap, dims)
        if got == -2:
            raise NotImplementedError(f"{path}: this JPEG process is not supported (baseline / extended sequential Huffman, 8 bit, "
                                      f"chroma at 1:1, 2:1 horizontally or 2:1 in both directions are)")
        h, w, c = dims
        if got < 0 or c != pix or h > rows or w > cols:
            raise ValueError(f"{path}: corrupt JPEG strip / tile (decoder returned {got}, frame {h} x {w} x {c} for a {rows} x {cols} x {pix} block)")
        return dst[:got].reshape(h, w, c)
    return dec


def _unpredict(block: np.ndarray, predictor: int, path) -> np.ndarray:
    """undo Predictor 2 (horizontal differencing per sample, TIFF 6.0 section 14) on a [rows, width, samples] block"""
    if predictor == 1:
        return block
    if predictor != 2 or block.dtype.kind == "f":
        raise NotImplementedError(f"{path}: TIFF predictor {predictor} on {block.dtype} is not supported")
    return np.cumsum(block, axis=1, dtype=block.dtype)


def _unpredict_float(raw: np.ndarray, rows: int, cols: int, pix: int, dt: np.dtype) -> np.ndarray:
    """undo Predictor 3 (floating-point horizontal differencing, Adobe Photoshop TIFF Technical Note 3; what GDAL writes with PREDICTOR=3):
    every row is stored as byte planes -- byte 0 (the most significant) of all its samples, then byte 1, ... -- independent of the file's byte
    order, and that byte row is differenced with a stride of `pix` bytes.  Returns [rows, cols, pix] in native byte order."""
    bps, n = dt.itemsize, cols * pix
    acc = np.cumsum(raw[:rows * n * bps].reshape(rows, n * bps // pix, pix), axis=1, dtype=np.uint8)          # (wraps modulo 256, as the bytes do)
    planes = acc.reshape(rows, bps, n)
    be = np.ascontiguousarray(np.moveaxis(planes, 1, 2))                                                          # [rows, n, bps] big-endian samples
    return be.view(np.dtype(f">f{bps}")).reshape(rows, cols, pix).astype(dt.newbyteorder("="))


def _geo_meta(tags) -> Dict:
    meta = {"tags": {t: tags[t] for t in GEO_TAGS if t in tags}, "geotransform": None}
    if 33550 in tags and 33922 in tags:
        sx, sy = tags[33550][0], tags[33550][1]
        i, j, _, x, y, _ = tags[33922][:6]
        meta["geotransform"] = (x - i * sx, sx, 0.0, y + j * sy, 0.0, -sy)
    if 42113 in tags:
        try:
            meta["nodata"] = float(tags[42113][0])
        except ValueError:
            meta["nodata"] = None
    return meta


def _one_error_contract(fn):
    """One error contract for the readers: a file this module does not implement raises NotImplementedError, a damaged one ValueError naming
    the file -- whatever the damage tripped inside (a missing tag, an offset past the end, a count that does not fit, zlib's own error, an
    allocation the header asks for and the file cannot justify).  GDAL, which the reference reads through, reports a read error there."""
    import functools
    import zlib

    @functools.wraps(fn)
    def reader(path):
        try:
            return fn(path)
        except (ValueError, NotImplementedError, OSError):
            raise
        except (KeyError, IndexError, struct.error, zlib.error, MemoryError, ArithmeticError, TypeError, UnicodeError) as e:
            raise ValueError(f"{path}: damaged TIFF ({type(e).__name__}: {e})") from e
    return reader


@_one_error_contract
def tiff_info(path) -> Dict:
    """Header-only read (memory-mapped: the pixel data is never touched): meta of read_tiff plus 'height', 'width', 'bands'.
    What the prediction merge needs from every tile before any of them is predicted (reference predict.py:206-222 takes the same
    numbers from gdal.Open(...).GetGeoTransform() / RasterXSize / RasterYSize)."""
    import mmap
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as b:
        tags, _ = _parse_tags(b, path)
    meta = _geo_meta(tags)
    meta.update(height=tags[257][0], width=tags[256][0], bands=tags.get(277, (1,))[0])
    return meta


@_one_error_contract
def read_tiff(path) -> Tuple[np.ndarray, Dict]:
    """Returns (array [C,H,W] (or [H,W] for one band), meta) with meta['geotransform'] = (ulx, xres, 0, uly, 0, -yres)
    when the file is georeferenced and meta['tags'] holding the raw GeoTIFF tags.  The file is memory-mapped (copy-on-write), never slurped:
    an uncompressed native-endian file with contiguous strips comes back as a VIEW of the mapping (no copy at all: the training feed's one
    copy is the one into its pinned staging buffer); everything else is decoded strip by strip into one output array."""
    import mmap
    import os
    with open(path, "rb") as f:
        if os.environ.get("UNET_TIFF_MMAP", "1") == "0":      # (A/B switch: read the file instead of mapping it)
            mm = bytearray(f.read())
        else:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)
    b = np.frombuffer(mm, dtype=np.uint8)           # (keeps the mapping alive for as long as a view of it exists)
    tags, bo = _parse_tags(mm, path)
    W, H = tags[256][0], tags[257][0]
    spp = tags.get(277, (1,))[0]
    bits = tags.get(258, (1,))[0]
    fmt = tags.get(339, (1,))[0]
    comp, predictor = tags.get(259, (1,))[0], tags.get(317, (1,))[0]
    planar = tags.get(284, (1,))[0]
    if bits not in (8, 16, 32, 64) or len(set(tags.get(258, (bits,)))) != 1:
        raise NotImplementedError(f"{path}: {tags.get(258)}-bit samples are not supported (8 / 16 / 32 / 64, the same for every band, are)")
    dt = _dtype(bits, fmt, bo)
    native = dt.newbyteorder("=")
    planes = spp if planar == 2 else 1
    pix = 1 if planar == 2 else spp
    meta = _geo_meta(tags)
    meta["dtype"] = native

    def finish(out):          # out: [planes, H, W, pix]
        arr = out[:, :, :, 0] if planar == 2 else np.moveaxis(out[0], -1, 0)
        return (arr[0] if arr.shape[0] == 1 else arr), meta

    if comp == 1 and predictor == 1 and 324 not in tags and dt.isnative:
        offs = tags[273]
        rps = min(tags.get(278, (H,))[0], H)
        total = planes * H * W * pix * dt.itemsize
        cnts = tags.get(279)
        contiguous = len(offs) == 1 or (cnts is not None and all(offs[k] + cnts[k] == offs[k + 1] for k in range(len(offs) - 1))
                                        and (planes == 1 or H % rps == 0))
        if contiguous and offs[0] + total <= b.size:
            return finish(b[offs[0]:offs[0] + total].view(dt).reshape(planes, H, W, pix))

    if comp == 7:
        if bits != 8 or predictor != 1:
            raise NotImplementedError(f"{path}: JPEG-in-TIFF with {bits}-bit samples / predictor {predictor} is not supported")
        jpeg = _jpeg_decoder(tags, path)
    decode = _decoder(comp, path) if comp != 7 else None
    # what the header asks for against what the file can hold: Deflate tops out at 1032 : 1, LZW and PackBits below that, a flat JPEG near
    # 100 : 1 -- a damaged ImageWidth / ImageLength (65535 x 65535 x 4 is 17 GB) is refused here instead of being allocated and decoded into
    if W <= 0 or H <= 0 or spp <= 0 or planes * H * W * pix * dt.itemsize > (1 << 20) + (b.size << (12 if comp != 1 else 0)):
        raise ValueError(f"{path}: {planes * pix} x {H} x {W} {dt} samples cannot come out of a file of {b.size} bytes (compression {comp})")
    out = np.zeros((planes, H, W, pix), dtype=native)

    def block(off, nbytes, rows, cols):
        """one strip / tile as [rows, cols, pix] in native byte order, decompressed and un-differenced"""
        if comp == 7:
            return jpeg(b[off:off + nbytes], rows, cols, pix)
        need = rows * cols * pix * dt.itemsize
        raw = decode(b[off:off + nbytes], need) if comp != 1 else b[off:off + need]
        if len(raw) < need:
            raise ValueError(f"{path}: strip / tile holds {len(raw)} bytes, {need} expected")
        if predictor == 3 and dt.kind == "f":
            return _unpredict_float(np.frombuffer(raw, dtype=np.uint8, count=need) if not isinstance(raw, np.ndarray) else raw[:need].view(np.uint8),
                                    rows, cols, pix, dt)
        t = raw[:need].view(dt).reshape(rows, cols, pix)
        return _unpredict(t.astype(native, copy=False), predictor, path)

    # every strip / tile is independent: a list of jobs (offset, bytes, block shape, destination), run by a few threads when the scene is
    # large (the codecs and zlib release the GIL) -- a 20000 x 20000 x 4 LZW scene is 1.6 GB to decode in front of a 5 s prediction
    jobs = []
    if 324 in tags:        # tiled
        tw, th = tags[322][0], tags[323][0]
        offs, cnts = tags[324], tags[325]
        tx, ty = -(-W // tw), -(-H // th)
        k = 0
        for p in range(planes):
            for j in range(ty):
                for i in range(tx):
                    h, w = min(th, H - j * th), min(tw, W - i * tw)
                    jobs.append((offs[k], cnts[k], th, tw, p, j * th, h, i * tw, w))
                    k += 1
    else:
        rps = min(tags.get(278, (H,))[0], H)
        offs = tags[273]
        spi = -(-H // rps)
        cnts = tags.get(279) or tuple(min(rps, H - (s % spi) * rps) * W * pix * dt.itemsize for s in range(planes * spi))
        for p in range(planes):
            for s_ in range(spi):
                r0 = s_ * rps
                rows = min(rps, H - r0)
                jobs.append((offs[p * spi + s_], cnts[p * spi + s_], rows, W, p, r0, rows, 0, W))

    def run(chunk):
        for off, nbytes, brows, bcols, p, r0, h, c0, w in chunk:
            t = block(off, nbytes, brows, bcols)
            if t.shape[0] < h or t.shape[1] < w:
                raise ValueError(f"{path}: strip / tile decodes to {t.shape[0]} x {t.shape[1]} samples, {h} x {w} needed")
            out[p, r0:r0 + h, c0:c0 + w] = t[:h, :w]

    total = planes * H * W * pix * dt.itemsize
    try:
        nthr = max(2, min(8, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        nthr = 4
    nthr = int(os.environ.get("UNET_TIFF_THREADS", nthr))
    if comp != 1 and len(jobs) >= 16 and total >= (32 << 20) and nthr > 1:
        from concurrent.futures import ThreadPoolExecutor
        step = max(1, -(-len(jobs) // (nthr * 8)))
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            list(ex.map(run, [jobs[i:i + step] for i in range(0, len(jobs), step)]))
    else:
        run(jobs)
    return finish(out)


def write_tiff(path, arr: np.ndarray, geotransform=None, tags: Optional[Dict[int, tuple]] = None, nodata=None, bigtiff: Optional[bool] = None) -> None:
    """Uncompressed, single strip, pixel-interleaved little-endian TIFF of a [C,H,W] or [H,W] array.  bigtiff None: BigTIFF (64-bit offsets)
    when the file would not fit 32-bit offsets -- what GDAL's BIGTIFF=IF_NEEDED does for the reference's outputs (predict.py:19-52): the
    all-classes probabilities of a 20000 x 20000 scene are 8 GB."""
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[None]
    C, H, W = a.shape
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype == np.int64:
        a = a.astype(np.int32)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    kind = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    nbytes = C * H * W * a.dtype.itemsize
    big = (nbytes > (1 << 32) - (1 << 20)) if bigtiff is None else bool(bigtiff)
    entries = []          # (tag, type, count, payload-bytes)
    off_t, off_fmt = (16, "Q") if big else (4, "I")          # LONG8 / LONG for offsets and byte counts

    def add(tag, typ, vals):
        if typ == 2:
            payload = vals.encode("latin1") + b"\0"
            cnt = len(payload)
        else:
            fmt = _TYPES[typ][0]
            payload = struct.pack("<" + fmt[0] * len(vals), *vals)
            cnt = len(vals)
        entries.append((tag, typ, cnt, payload))

    add(256, 4, [W]); add(257, 4, [H]); add(258, 3, [a.dtype.itemsize * 8] * C); add(259, 3, [1])
    add(262, 3, [1]); add(273, off_t, [0]); add(277, 3, [C]); add(278, 4, [H]); add(279, off_t, [nbytes]); add(284, 3, [1])
    if C > 1:
        add(338, 3, [0] * (C - 1))      # ExtraSamples: unspecified (what GDAL writes for MINISBLACK multi-band)
    add(339, 3, [kind] * C)
    geo = dict(tags or {})
    if geotransform is not None:
        ulx, xres, _, uly, _, yres = geotransform
        geo[33550] = (abs(xres), abs(yres), 0.0)
        geo[33922] = (0.0, 0.0, 0.0, ulx, uly, 0.0)
    if nodata is not None:
        geo[42113] = (str(nodata),)
    for t in sorted(geo):
        v = geo[t]
        if t in (34737, 42113):
            add(t, 2, v[0] if isinstance(v, tuple) else v)
        elif t == 34735:
            add(t, 3, list(v))
        else:
            add(t, 12, list(v))
    entries.sort(key=lambda e: e[0])
    n = len(entries)
    inl = 8 if big else 4                                     # bytes of the inline value / offset field
    ifd_off = 16 if big else 8
    extra_off = ifd_off + (8 + 20 * n + 8 if big else 2 + 12 * n + 4)
    extra = b""
    placed = []
    for tag, typ, cnt, payload in entries:
        if len(payload) <= inl:
            placed.append((tag, typ, cnt, payload.ljust(inl, b"\0")))
        else:
            if len(extra) % 2:
                extra += b"\0"
            placed.append((tag, typ, cnt, struct.pack("<" + off_fmt, extra_off + len(extra))))
            extra += payload
    if len(extra) % 2:
        extra += b"\0"
    data_off = extra_off + len(extra)
    body = b""
    for tag, typ, cnt, val in placed:
        if tag == 273:
            val = struct.pack("<" + off_fmt, data_off)
        body += struct.pack("<HH" + ("Q" if big else "I"), tag, typ, cnt) + val
    if big:
        head = b"II" + struct.pack("<HHHQ", 43, 8, 0, ifd_off) + struct.pack("<Q", n) + body + struct.pack("<Q", 0)
    else:
        head = b"II" + struct.pack("<HI", 42, ifd_off) + struct.pack("<H", n) + body + struct.pack("<I", 0)
    with open(path, "wb") as f:                               # the pixel data is streamed band-interleaved row block by row block: no second copy
        f.write(head + extra)
        le = a.dtype.newbyteorder("<")
        rows = max(1, (64 << 20) // max(1, C * W * a.dtype.itemsize))
        for r0 in range(0, H, rows):
            f.write(np.ascontiguousarray(np.moveaxis(a[:, r0:r0 + rows], 0, -1)).astype(le, copy=False).tobytes())
