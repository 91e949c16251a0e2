"""Golden files of the GeoTIFF reader's codecs, written by an INDEPENDENT implementation: Pillow's libtiff binding (PIL 12.2, libtiff) in
this container.  The arrays are regenerated from the seed by tests/test_tiling_cpu.py; the files are data, committed.
    python tests/golden/make_tiff_golden.py
Why these: GDAL (which the reference writes and reads rasters with: predict.py:19-52, create_tiles_unet.py:252-434) produces LZW or Deflate
strips, optionally with PREDICTOR=2, and BigTIFF beyond 4 GB."""
import os
import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiff")


def arrays():
    rng = np.random.default_rng(20261005)
    smooth = (np.add.outer(np.arange(97), np.arange(131)) % 256).astype(np.uint8)          # long runs / repeats: deep LZW tables, KwKwK codes
    return {"u8": np.where(rng.random((97, 131)) < 0.1, rng.integers(0, 256, (97, 131)), smooth).astype(np.uint8),
            "rgba": (rng.integers(0, 256, (60, 70, 4)) // 32 * 32).astype(np.uint8),
            "u16": rng.integers(0, 4000, (50, 45)).astype(np.uint16),
            "big": (rng.integers(0, 8, (300, 517)) * 31).astype(np.uint8)}                 # > 4096 table entries per strip: ClearCode inside a strip


if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    A = arrays()
    for name, comp, pred, bigtiff in [("u8", "tiff_lzw", False, False), ("u8", "tiff_lzw", True, False), ("rgba", "tiff_lzw", True, False),
                                      ("rgba", "tiff_adobe_deflate", False, False), ("u16", "tiff_adobe_deflate", True, False),
                                      ("u16", "tiff_lzw", True, False), ("u8", "packbits", False, False), ("big", "tiff_lzw", False, False),
                                      ("u16", "tiff_lzw", False, True), ("rgba", None, False, True)]:
        kw = {"compression": comp} if comp else {}
        if pred:
            kw["tiffinfo"] = {317: 2}
        if bigtiff:
            kw["big_tiff"] = True
        f = os.path.join(HERE, f"{name}_{comp or 'none'}{'_pred2' if pred else ''}{'_bigtiff' if bigtiff else ''}.tif")
        Image.fromarray(A[name]).save(f, format="TIFF", **kw)
        print(f, os.path.getsize(f))
