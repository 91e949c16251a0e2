"""Raster tiling of the tile workflow (reference ``create_tiles_unet.py``: ``compute_windows`` :30-56, ``split_raster``
:252-434, ``save_crop`` :179-249, ``create_train_test_split`` :69-176) without GDAL / rasterio / slidingwindow: plain numpy +
``unet_amd.tiffio``.  CPU-side preprocessing -- not part of the GPU hot path.

Window rule (``slidingwindow.generate`` 0.0.14 as the reference calls it): step = patch_size - floor(patch_size * overlap);
offsets 0, step, ... while the window fits, plus one window flush with the far edge when the raster is not covered
(20000 px, 512 px windows, overlap 0.2 -> 0, 410, ..., 19270, 19488: 49 per axis), row-major order.
"""
from __future__ import annotations

import warnings
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

from unet_amd.mosaic import window_offsets
from unet_amd.tiffio import read_tiff, write_tiff


def compute_windows(height: int, width: int, patch_size: int, patch_overlap: float) -> List[Tuple[int, int, int, int]]:
    """[(x, y, w, h)] windows over a height x width raster (the rule itself: unet_amd.mosaic.sliding_windows, shared with predict_raster)."""
    if patch_overlap > 1:
        raise ValueError(f"Patch overlap {patch_overlap} must be between 0 - 1")
    step = patch_size - int(patch_size * patch_overlap)
    return [(x, y, patch_size, patch_size) for y in window_offsets(height, patch_size, step) for x in window_offsets(width, patch_size, step)]


def create_train_test_split(path, split=None, seed: Optional[int] = None):
    """Moves the tiles under <path>/img_tiles and <path>/mask_tiles into trai/ vali/ [test/] by a random shuffle."""
    if split is None:
        split = [0.7, 0.2, 0.1]
    if np.round(np.sum(split), decimals=3) != 1.0:
        split = [0.7, 0.2, 0.1]
        warnings.warn("Train/Vali/Test-Split percentage does not sum to 1, reseting to 70%/20%/10%.")
    root = Path(path)
    names = sorted(p.name for p in (root / "img_tiles").glob("*.tif"))
    rng = np.random.default_rng(seed)
    rng.shuffle(names)
    n = len(names)
    n_tr = int(n * split[0])
    three = len(split) == 3 and split[-1] != 0
    n_va = int(n * np.sum(split[:2])) if three else n
    parts = {"trai": names[:n_tr], "vali": names[n_tr:n_va]}
    if three:
        parts["test"] = names[n_va:]
    for part, files in parts.items():
        for kind in ("img_tiles", "mask_tiles"):
            (root / part / kind).mkdir(parents=True, exist_ok=True)
            for f in files:
                src = root / kind / f
                if src.exists():
                    src.rename(root / part / kind / f)
    for kind in ("img_tiles", "mask_tiles"):
        d = root / kind
        if d.exists() and not any(d.iterdir()):
            d.rmdir()
    return {k: len(v) for k, v in parts.items()}


def split_raster(path_to_raster=None, path_to_mask=None, base_dir=".", patch_size=400, patch_overlap=0.20, split=None, max_empty=0.9,
                 class_zero=False, seed: Optional[int] = None):
    """Cut a raster (and its mask) into patch_size tiles, drop tiles that are more than max_empty empty, write GeoTIFF tiles
    with their own geotransform and split them into trai / vali / test."""
    if split is None:
        split = [0.7, 0.2, 0.1]
    base = Path(base_dir)
    img, meta = read_tiff(path_to_raster)
    img = img[None] if img.ndim == 2 else img
    gt = meta["geotransform"] or (0.0, 1.0, 0.0, 0.0, 0.0, -1.0)
    nodata = meta.get("nodata")
    mask = None
    if path_to_mask is not None:
        mask, mmeta = read_tiff(path_to_mask)
        mask = mask[None] if mask.ndim == 2 else mask
        mgt = mmeta["geotransform"] or gt
        if mask.shape[1:] != img.shape[1:] or mgt[0] != gt[0] or mgt[3] != gt[3]:
            # use the overlapping area (assumes the same pixel size, as the reference does)
            x0, y0 = max(gt[0], mgt[0]), min(gt[3], mgt[3])
            x1 = min(gt[0] + img.shape[2] * gt[1], mgt[0] + mask.shape[2] * mgt[1])
            y1 = max(gt[3] + img.shape[1] * gt[5], mgt[3] + mask.shape[1] * mgt[5])
            w, h = int(round((x1 - x0) / gt[1])), int(round((y1 - y0) / gt[5]))
            ix, iy = int(round((x0 - gt[0]) / gt[1])), int(round((y0 - gt[3]) / gt[5]))
            mx, my = int(round((x0 - mgt[0]) / mgt[1])), int(round((y0 - mgt[3]) / mgt[5]))
            img, mask = img[:, iy:iy + h, ix:ix + w], mask[:, my:my + h, mx:mx + w]
            gt = (x0, gt[1], 0.0, y0, 0.0, gt[5])
        if class_zero:
            mask = mask + 1        # class ids shift by one so that 0 stays the nodata / background marker
        mnodata = mmeta.get("nodata")
        bad = np.zeros(img.shape[1:], dtype=bool)
        if nodata is not None:
            bad |= (img == nodata).any(axis=0)
        if mnodata is not None:
            bad |= (mask == (mnodata + 1 if class_zero else mnodata)).any(axis=0)
        img = img.copy(); mask = mask.copy()
        img[:, bad] = 0
        mask[:, bad] = 0
    elif nodata is not None:
        img = img.copy()
        img[:, (img == nodata).any(axis=0)] = 0
    H, W = img.shape[1:]
    if H < patch_size or W < patch_size:
        raise ValueError(f"Patch size of {patch_size} is larger than the image dimensions {[H, W]}")
    name = Path(path_to_raster).stem
    (base / "img_tiles").mkdir(parents=True, exist_ok=True)
    if mask is not None:
        (base / "mask_tiles").mkdir(parents=True, exist_ok=True)
    kept = 0
    for index, (x, y, w, h) in enumerate(compute_windows(H, W, patch_size, patch_overlap)):
        crop = img[:, y:y + h, x:x + w]
        if crop.size == 0 or np.sum(crop != 0) < crop.size * (1 - max_empty):
            continue
        if mask is not None:
            mcrop = mask[:, y:y + h, x:x + w]
            if mcrop.size == 0 or np.sum(mcrop != 0) < mcrop.size * (1 - max_empty):
                continue
        tgt = (gt[0] + x * gt[1], gt[1], 0.0, gt[3] + y * gt[5], 0.0, gt[5])
        write_tiff(base / "img_tiles" / f"{name}_{index}.tif", crop, geotransform=tgt, tags=meta["tags"])
        if mask is not None:
            md = mcrop[0] if mcrop.dtype.kind == "f" else mcrop[0].astype(np.uint8)
            write_tiff(base / "mask_tiles" / f"{name}_{index}.tif", md, geotransform=tgt, tags=meta["tags"])
        kept += 1
    counts = create_train_test_split(base, split=split, seed=seed) if mask is not None else {"tiles": kept}
    return counts
