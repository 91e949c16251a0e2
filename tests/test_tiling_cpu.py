"""CPU suite: raster tiling (window rule, emptiness filter, geotransforms, split) and the flip augmentation slicing rule."""
import numpy as np
import torch

import create_tiles_unet as T
from unet_amd.learner import FlipAugment
from unet_amd.tiffio import read_tiff, write_tiff


def test_window_rule_matches_survey_example():
    ws = T.compute_windows(20000, 20000, 512, 0.2)
    xs = sorted({w[0] for w in ws})
    assert len(ws) == 49 * 49 and xs[:3] == [0, 410, 820] and xs[-2:] == [19270, 19488]
    assert T.compute_windows(800, 1200, 400, 0.0) == [(x, y, 400, 400) for y in (0, 400) for x in (0, 400, 800)]
    assert [w[0] for w in T.compute_windows(400, 1000, 400, 0.0)] == [0, 400, 600]             # last window flush with the edge


def test_split_raster_end_to_end(tmp_path):
    g = np.random.default_rng(0)
    img = g.integers(1, 255, (4, 300, 420)).astype(np.uint8)
    mask = g.integers(1, 4, (300, 420)).astype(np.uint8)
    img[:, :100, :100] = 0            # an empty corner: that tile must be dropped
    mask[:100, :100] = 0
    gt = (1000.0, 0.5, 0.0, 2000.0, 0.0, -0.5)
    write_tiff(tmp_path / "r.tif", img, geotransform=gt)
    write_tiff(tmp_path / "m.tif", mask, geotransform=gt)
    counts = T.split_raster(tmp_path / "r.tif", tmp_path / "m.tif", tmp_path / "out", patch_size=100, patch_overlap=0.0,
                            split=[0.75, 0.25], max_empty=0.9, seed=1)
    n_windows = len(T.compute_windows(300, 420, 100, 0.0))
    assert n_windows == 15 and counts["trai"] + counts["vali"] == 14 and counts["trai"] == 10
    tiles = sorted((tmp_path / "out" / "trai" / "img_tiles").glob("*.tif"))
    t, meta = read_tiff(tiles[0])
    idx = int(tiles[0].stem.split("_")[-1])
    x, y, _, _ = T.compute_windows(300, 420, 100, 0.0)[idx]
    assert np.array_equal(t, img[:, y:y + 100, x:x + 100])
    assert meta["geotransform"] == (1000.0 + x * 0.5, 0.5, 0.0, 2000.0 - y * 0.5, 0.0, -0.5)
    m, _ = read_tiff(tmp_path / "out" / "trai" / "mask_tiles" / tiles[0].name)
    assert np.array_equal(m, mask[y:y + 100, x:x + 100])


def test_flip_augment_slicing_rule():
    x = torch.arange(4 * 1 * 2 * 3, dtype=torch.float32).view(4, 1, 2, 3)
    y = torch.arange(4 * 2 * 3).view(4, 2, 3)
    a, b = FlipAugment(1.0, 0.0, n_transform_imgs=1.0)(x.clone(), y.clone())
    assert torch.equal(a, x) and torch.equal(b, y)                  # quirk Q7: default touches nothing
    a, b = FlipAugment(1.0, 0.0, n_transform_imgs=0.5)(x.clone(), y.clone())
    assert torch.equal(a[:2], x[:2].flip(-1)) and torch.equal(a[2:], x[2:]) and torch.equal(b[:2], y[:2].flip(-1))


def test_compressed_and_bigtiff_golden_files_written_by_libtiff():
    """LZW / Deflate / PackBits strips, Predictor 2, BigTIFF: the files under tests/golden/tiff were written by Pillow's libtiff binding
    (tests/golden/make_tiff_golden.py) -- an independent implementation -- from seeded arrays; the reader must return those arrays bit for
    bit.  This is what lets predict_raster("scene.tif") open the rasters GDAL writes (reference create_tiles_unet.py:252-434 reads through
    gdal.Open, data.py:18-28 through rasterio: both decode these transparently)."""
    import glob
    import importlib.util
    import os
    from unet_amd.tiffio import tiff_info
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_tiff_golden", os.path.join(here, "make_tiff_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    A = mod.arrays()
    files = sorted(glob.glob(os.path.join(here, "tiff", "*.tif")))
    assert len(files) >= 10
    seen = set()
    for f in files:
        name = os.path.basename(f).split("_")[0]
        ref = A[name] if A[name].ndim == 2 else np.moveaxis(A[name], -1, 0)
        got, meta = read_tiff(f)
        assert got.dtype == ref.dtype and np.array_equal(got, ref), f
        info = tiff_info(f)
        assert (info["height"], info["width"]) == ref.shape[-2:], f
        seen.add(open(f, "rb").read(4)[2])
    assert seen == {42, 43}          # classic and BigTIFF headers


def test_lzw_decoder_rejects_garbage_and_overflow():
    import ctypes as C
    from unet_amd._lib import lib
    dst = C.create_string_buffer(16)
    assert lib.unet_tiff_lzw_decode(bytes([0xff] * 8), 8, dst, 16) == -1            # a code beyond the table
    # ClearCode, 'A', 'B', EOI as 9-bit MSB-first codes: 100000000 001000001 001000010 100000001
    bits = "100000000" + "001000001" + "001000010" + "100000001"
    bits += "0" * (-len(bits) % 8)
    src = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    assert lib.unet_tiff_lzw_decode(src, len(src), dst, 16) == 2 and dst.raw[:2] == b"AB"
    assert lib.unet_tiff_lzw_decode(src, len(src), dst, 1) == -1                     # capacity exceeded
    assert lib.unet_tiff_packbits_decode(bytes([0x02, 1, 2, 3, 0xfe, 9]), 6, dst, 16) == 6 and dst.raw[:6] == bytes([1, 2, 3, 9, 9, 9])


def test_tiff_decoders_against_libtiff_on_random_images(tmp_path):
    """fuzz: random sizes / contents / sample types written by Pillow's libtiff binding with LZW, Deflate and PackBits (with and without
    Predictor 2) must read back bit for bit -- high-entropy uint8 strips fill the 4096-entry LZW table (ClearCode inside a strip), the
    run-length images exercise long KwKwK chains"""
    from PIL import Image
    rng = np.random.default_rng(20261005)
    n = 0
    for trial in range(24):
        H, W = int(rng.integers(1, 260)), int(rng.integers(1, 260))
        kind = trial % 4
        if kind == 0:
            a = rng.integers(0, 256, (H, W)).astype(np.uint8)
        elif kind == 1:
            a = (rng.integers(0, 4, (H, W)) * 60).astype(np.uint8)
        elif kind == 2:
            a = rng.integers(0, 65535, (H, W)).astype(np.uint16)
        else:
            a = (np.add.outer(np.arange(H), np.arange(W)) % 7).astype(np.uint8)[..., None].repeat(3, 2)
        ref = a if a.ndim == 2 else np.moveaxis(a, -1, 0)
        for comp in ("tiff_lzw", "tiff_adobe_deflate", "packbits"):
            for pred in (False, True):
                if pred and comp == "packbits":
                    continue
                kw = {"compression": comp}
                if pred:
                    kw["tiffinfo"] = {317: 2}
                Image.fromarray(a).save(tmp_path / "t.tif", format="TIFF", **kw)
                got, _ = read_tiff(tmp_path / "t.tif")
                assert np.array_equal(got, ref), (trial, comp, pred, a.shape)
                n += 1
    assert n == 120


def test_float_predictor_3_against_libtiff(tmp_path):
    """Predictor 3 (floating-point horizontal differencing: what GDAL writes for float rasters with PREDICTOR=3, e.g. the probability rasters of
    reference predict.py:19-52 re-compressed): files written by Pillow's libtiff binding with LZW / Deflate read back bit for bit (Pillow writes
    one float band; the multi-sample stride is the next test's)"""
    from PIL import Image
    rng = np.random.default_rng(7)
    n = 0
    for trial in range(10):
        H, W = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        a = (rng.normal(size=(H, W)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        if trial % 3 == 0:
            a[rng.integers(0, H), rng.integers(0, W)] = np.float32("inf")
        for comp in ("tiff_lzw", "tiff_adobe_deflate"):
            Image.fromarray(a).save(tmp_path / "f.tif", format="TIFF", compression=comp, tiffinfo={317: 3})
            got, meta = read_tiff(tmp_path / "f.tif")
            assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), a.view(np.uint32)), (trial, comp, a.shape)
            n += 1
    assert n == 20


def test_float_predictor_3_multi_sample_rows():
    """the byte differencing of Predictor 3 runs with a stride of samples-per-pixel bytes over the byte-plane row: checked against the encoder
    of the specification (Adobe TIFF Technical Note 3) restated here, float32 and float64, 1 / 3 / 4 samples per pixel"""
    from unet_amd.tiffio import _unpredict_float
    rng = np.random.default_rng(11)
    for dt in (np.dtype("<f4"), np.dtype("<f8")):
        for pix in (1, 3, 4):
            rows, cols = 5, 9
            a = rng.normal(size=(rows, cols, pix)).astype(dt)
            be = a.astype(dt.newbyteorder(">")).view(np.uint8).reshape(rows, cols * pix, dt.itemsize)
            planes = np.moveaxis(be, 2, 1).reshape(rows, -1).astype(np.int16)                # byte 0 of every sample, then byte 1, ...
            enc = planes.copy()
            enc[:, pix:] = planes[:, pix:] - planes[:, :-pix]
            raw = (enc % 256).astype(np.uint8).reshape(-1)
            got = _unpredict_float(raw, rows, cols, pix, dt)
            assert np.array_equal(got, a), (dt, pix)


def test_write_tiff_switches_to_bigtiff_and_both_forms_are_read_by_libtiff(tmp_path):
    """write_tiff emits BigTIFF when the data would not fit 32-bit offsets (GDAL's BIGTIFF=IF_NEEDED: the all-classes probabilities of a
    20000 x 20000 scene are 8 GB); both header forms are read back by this reader and by Pillow's libtiff (single-band files: Pillow has
    no mode for 3 MINISBLACK samples)"""
    from PIL import Image
    rng = np.random.default_rng(3)
    gt = (400000.0, 0.5, 0.0, 5700000.0, 0.0, -0.5)
    for big in (False, True):
        a = rng.integers(0, 255, (3, 37, 53)).astype(np.uint8)
        write_tiff(tmp_path / "w.tif", a, geotransform=gt, nodata=0, bigtiff=big)
        assert open(tmp_path / "w.tif", "rb").read(4) == (b"II+\x00" if big else b"II*\x00")
        b, meta = read_tiff(tmp_path / "w.tif")
        assert np.array_equal(a, b) and meta["geotransform"] == gt and meta["nodata"] == 0.0
        for m in (rng.integers(0, 255, (41, 29)).astype(np.uint8), rng.random((41, 29)).astype(np.float32)):
            write_tiff(tmp_path / "m.tif", m, geotransform=gt, bigtiff=big)
            assert np.array_equal(np.array(Image.open(tmp_path / "m.tif")), m)
            g, _ = read_tiff(tmp_path / "m.tif")
            assert np.array_equal(g, m)
    # the automatic switch: decided from the byte count alone (not written here: 4 GB)
    import inspect
    from unet_amd import tiffio
    assert "(1 << 32) - (1 << 20)" in inspect.getsource(tiffio.write_tiff)
