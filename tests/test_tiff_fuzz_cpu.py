"""Damaged strips through the host TIFF decoders under AddressSanitizer / UBSan (CPU build only: tests/fuzz/tiff_fuzz.cpp): a corrupt
tile file has to end in an error code -- tiffio.read_tiff raises ValueError -- never in an access outside the decoder's buffers.  Streams
come from libtiff / libjpeg (Pillow); each is first decoded intact, then a few thousand times with cuts, overwritten bytes, 0xFF runs,
deleted spans and single flipped bits, into full-size and too-small destination buffers."""
import io
import os
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

Image = pytest.importorskip("PIL.Image")
REPO = Path(__file__).resolve().parent.parent


def _scene(h, w, c, rng):
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([np.sin(x / (5 + i)) * 60 + np.cos(y / (7 + i)) * 60 + 128 for i in range(c)], -1) + rng.normal(0, 12, (h, w, c))
    return np.clip(a, 0, 255).astype(np.uint8)


def _strips(path):
    """(JPEGTables or b'', [strip bytes]) of a libtiff-written file."""
    with Image.open(path) as im:
        tags = im.tag_v2
        offs, cnts = tags[273], tags[279]
        tables = bytes(tags[347]) if 347 in tags else b""
    raw = Path(path).read_bytes()
    return tables, [raw[o:o + n] for o, n in zip(offs, cnts)]


def _corpus(tmp_path):
    rng = np.random.default_rng(5)
    recs = []
    modes = {1: "L", 3: "RGB", 4: "CMYK"}
    for c, kw in ((3, dict(quality=85, subsampling=0)), (3, dict(quality=85, subsampling=1)), (3, dict(quality=70, subsampling=2)),
                  (1, dict(quality=90)), (4, dict(quality=90)), (3, dict(quality=80, subsampling=2, optimize=True)),
                  (3, dict(quality=80, restart_marker_blocks=2))):
        for h, w in ((40, 56), (33, 47)):
            im = _scene(h, w, c, rng)
            b = io.BytesIO()
            Image.fromarray(im[..., 0] if c == 1 else im, modes[c]).save(b, "JPEG", **kw)
            recs.append((0, b"", b.getvalue(), h * w * c))
    for c, mode in ((3, "RGB"), (4, "RGBA"), (1, "L")):                  # abbreviated streams + JPEGTables, as libtiff writes them
        im = _scene(48, 64, c, rng)
        p = tmp_path / f"j{c}.tif"
        Image.fromarray(im[..., 0] if c == 1 else im, mode).save(p, compression="jpeg", quality=85)
        tables, strips = _strips(p)
        assert tables and len(strips) == 1
        recs.append((1, tables, strips[0], 48 * 64 * c))
    for kind, comp in ((2, "tiff_lzw"), (3, "packbits")):
        for k, im in enumerate((_scene(64, 96, 3, rng), np.repeat(_scene(8, 96, 3, rng), 8, axis=0), rng.integers(0, 256, (32, 64, 3), dtype=np.uint8))):
            p = tmp_path / f"{comp}{k}.tif"
            Image.fromarray(im, "RGB").save(p, compression=comp)
            _, strips = _strips(p)
            rows = im.shape[0] // len(strips) if im.shape[0] % len(strips) == 0 else None
            if rows is None:
                continue
            for s in strips[:2]:
                recs.append((kind, b"", s, rows * im.shape[1] * 3))
    out = tmp_path / "corpus.bin"
    with open(out, "wb") as f:
        for kind, tables, stream, size in recs:
            f.write(struct.pack("<BI", kind, len(tables)) + tables + struct.pack("<I", len(stream)) + stream + struct.pack("<I", size))
    return out, len(recs)


def test_damaged_strips_never_leave_the_decoders_buffers(tmp_path):
    cxx = os.environ.get("CXX", "g++")
    if shutil.which(cxx) is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "tiff_fuzz"
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                        str(REPO / "tests" / "fuzz" / "tiff_fuzz.cpp"), "-x", "c++", str(REPO / "unet_amd" / "csrc" / "tiff_codecs.hip"),
                        str(REPO / "unet_amd" / "csrc" / "host" / "tiff_jpeg.cpp"), "-I", str(REPO / "include"), "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr + r.stdout or r.returncode != 0 and "asan" in (r.stderr + r.stdout).lower():
        pytest.skip("this compiler has no sanitizer runtime: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    corpus, n = _corpus(tmp_path)
    assert n >= 20
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2):
        r = subprocess.run([str(exe), str(corpus), "1500", str(seed)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        words = r.stdout.split()
        got = dict(zip(words[0::2], map(int, words[1::2])))
        assert got["streams"] == n and got["intact"] == n          # every stream decodes to its full size before it is damaged
        assert got["refused"] > 0 and got["decoded"] > 0, r.stdout


def test_damaged_files_raise_value_error_and_nothing_else(tmp_path):
    """Whole files (header, tags, strips) damaged at random through tiffio.read_tiff / tiff_info: the only outcomes are an array, ValueError
    naming the file, or NotImplementedError -- no KeyError / struct.error / zlib.error / MemoryError from inside, and no read that takes
    seconds because a damaged count or image size was believed (both were found by this loop before the checks went in)."""
    import time
    from unet_amd.tiffio import read_tiff, tiff_info, write_tiff
    rng = np.random.default_rng(11)
    im = _scene(48, 64, 4, rng)
    files = {}
    write_tiff(tmp_path / "u.tif", np.moveaxis(im, -1, 0), geotransform=(10.0, 0.5, 0.0, 20.0, 0.0, -0.5))
    files["none"] = (tmp_path / "u.tif").read_bytes()
    for comp in ("tiff_lzw", "jpeg", "packbits", "tiff_adobe_deflate"):
        Image.fromarray(im, "RGBA").save(tmp_path / "c.tif", compression=comp)
        files[comp] = (tmp_path / "c.tif").read_bytes()
    Image.fromarray(im[..., :3], "RGB").save(tmp_path / "c.tif", compression="jpeg")
    files["jpeg_ycc"] = (tmp_path / "c.tif").read_bytes()
    names = list(files)
    seen = {"ok": 0, "refused": 0}
    target = tmp_path / "m.tif"
    for it in range(4000):
        b = bytearray(files[names[it % len(names)]])
        m = int(rng.integers(4))
        if m == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(len(b)))] = int(rng.integers(256))
        elif m == 1:                                  # the tag area: libtiff / Pillow put the IFD at the end, write_tiff at the start
            lo = max(0, len(b) - 400) if rng.integers(2) else 0
            for _ in range(int(rng.integers(1, 4))):
                b[lo + int(rng.integers(min(400, len(b) - lo)))] = int(rng.integers(256))
        elif m == 2:
            b = b[:int(rng.integers(len(b)))]
        else:
            a = int(rng.integers(len(b)))
            b[a:a + 4] = b"\xff\xff\xff\xff"[:len(b) - a]
        target.write_bytes(bytes(b))
        t0 = time.perf_counter()
        for fn in (tiff_info, read_tiff):
            try:
                fn(target)
                seen["ok"] += 1
            except (ValueError, NotImplementedError) as e:
                assert isinstance(e, NotImplementedError) or str(target) in str(e) or "mmap" in str(e) or "empty" in str(e), e
                seen["refused"] += 1
        assert time.perf_counter() - t0 < 2.0, (names[it % len(names)], m)
    assert seen["ok"] > 500 and seen["refused"] > 500, seen
