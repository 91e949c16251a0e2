"""read_tiff of a compressed 8000 x 8000 x 4 scene (libtiff-written LZW / Deflate): strips decoded by a thread pool against one thread"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pathlib import Path
from PIL import Image
from unet_amd.tiffio import read_tiff
d = Path(tempfile.mkdtemp()); g = np.random.default_rng(0)
H = 8000
sm = ((np.add.outer(np.arange(H), np.arange(H)) // 8)[..., None] + np.arange(4) * 7 + g.integers(0, 3, (H, H, 4))).astype(np.uint8)
for comp in ("tiff_lzw", "tiff_adobe_deflate"):
    Image.fromarray(sm, "RGBA").save(d / "big.tif", compression=comp)
    for thr in ("1", "8", "1", "8"):
        os.environ["UNET_TIFF_THREADS"] = thr
        t = time.perf_counter(); a, _ = read_tiff(d / "big.tif"); dt = time.perf_counter() - t
        print(comp, os.path.getsize(d / "big.tif") >> 20, "MB file, threads", thr, f"{dt:.2f} s {sm.nbytes / dt / 1e6:.0f} MB/s", bool(np.array_equal(a[:, :64], np.moveaxis(sm[:64], -1, 0))), flush=True)
