"""Prediction entry point of the tile workflow on the MI355X hot path.

Mirrors ``save_predictions`` (reference ``predict.py:146-147``).  Differences that keep results identical but not the
schedule: tiles are predicted in BATCHES (the reference loops ``learn.predict`` one tile at a time, ``predict.py:191-193``) and the
overlap merge -- sum of softmax probabilities + hit counter -> divide -> argmax (``predict.py:284-334``) -- runs on the GPU
(``unet_mosaic_accumulate`` / ``unet_mosaic_finalize``).  ``regression`` predicts the raw single-band output (float32 tiles; merged
mosaic = mean of the overlapping tiles, nodata -9999 where no tile was placed).  The confusion-matrix plots are out of scope.
"""
from __future__ import annotations

import os
import time
import warnings
from pathlib import Path

import numpy as np
import torch

from unet_amd import ops
from unet_amd.learner import load_learner, open_tile, scale_input
from unet_amd.tiffio import read_tiff, write_tiff


def store_tif(output_file, data, geotrans=None, tags=None, nodata=None, class_zero=False):
    """predict.py:19-52: GeoTIFF writer; class_zero shifts class ids back by one (0 was reserved for nodata)."""
    a = np.asarray(data)
    if class_zero and a.dtype.kind in "ui":
        a = a - 1 if a.dtype.kind == "i" else (a.astype(np.int16) - 1)
    write_tiff(output_file, a, geotransform=geotrans, tags=tags, nodata=nodata)


def _geo(path):
    if Path(path).suffix == ".npy":
        return None, {}
    _, meta = read_tiff(path)
    return meta["geotransform"], meta["tags"]


def save_predictions(predict_model, predict_path, regression, merge=False, all_classes=False, specific_class=None, large_file=False,
                     AOI=None, year=None, validation_vision=True, class_zero=False, batch_size=16):
    # cfg5: one process per GPU (torch.distributed.run); tile i goes to rank i mod world, the merge mosaic is summed with RCCL
    from unet_amd.distributed import init_from_env
    import torch.distributed as dist
    rank, local_rank, world = init_from_env()
    learn = load_learner(Path(predict_model), device=f"cuda:{local_rank}" if world > 1 else "cuda")
    model = learn.model
    path = Path(predict_path)
    output_folder = path.parent if merge else path.parent / ("predicted_tiles_" + Path(predict_model).stem)
    output_folder.mkdir(parents=True, exist_ok=True)
    model_name = os.path.basename(predict_model).split(".")[0]
    tiles = sorted([p for p in path.iterdir() if p.suffix.lower() in (".tif", ".tiff", ".npy")])
    print(f"Started at: {time.strftime('%H:%M:%S')}  ({len(tiles)} tiles)")
    dtype = learn.dls.train_ds.dtype
    geos = [_geo(t) for t in tiles]
    if merge:
        gts = np.array([[g[0][0], 0, g[0][1], g[0][3], 0, g[0][5]] for g in geos], dtype=np.float64)
    results = []          # (tile index, probs [C,H,W] device tensor) when merging
    mosaic = count = None
    mine = list(range(rank, len(tiles), world))
    for b0 in range(0, len(mine), batch_size):
        ids = mine[b0:b0 + batch_size]
        chunk = [tiles[i] for i in ids]
        x = torch.from_numpy(np.stack([scale_input(open_tile(t), dtype) for t in chunk]))
        if regression:       # predict.py:195-197: tile_preds[1] = raw outputs [1,H,W]
            probs, amax = model.predict_values(x.to(model._device)), None
        else:
            probs, amax = model.predict_probs(x.to(model._device))
        for j, t in enumerate(chunk):
            i = ids[j]
            gt, tags = geos[i]
            if merge:
                results.append((i, probs[j]))
                gts[i, 1], gts[i, 4] = probs.shape[3], probs.shape[2]
                continue
            if regression or all_classes:
                out = probs[j].cpu().numpy()
            elif specific_class is None:
                out = amax[j].cpu().numpy().astype(np.uint8)
            else:
                out = probs[j, specific_class].cpu().numpy()
            if large_file and out.dtype.kind == "f" and out.max() <= 1 and (all_classes or specific_class):
                out = np.around(out * ((128 / 4) - 1)).astype(np.int8)
            name = t.name if t.suffix != ".npy" else t.stem + ".tif"
            store_tif(output_folder / name, out, gt, tags, None, class_zero)
    if validation_vision:
        pass  # per-tile majority-class confusion plots (predict.py:56-143) are reporting, out of scope
    if not merge:
        if world > 1:
            dist.barrier()
        print(f"Prediction stored in {output_folder}.")
        return output_folder
    if world > 1:       # every rank needs every tile's size for the mosaic extent
        sizes = torch.from_numpy(gts[:, [1, 4]].copy()).to(model._device)
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX)
        gts[:, [1, 4]] = sizes.cpu().numpy()
    # ---- overlap merge (predict.py:257-355): mosaic extent from the tiles' geotransforms
    ulx_full, uly_full = gts[:, 0].min(), gts[:, 3].max()
    xres, yres = gts[0, 2], gts[0, 5]
    xmax_i, ymin_i = gts[:, 0].argmax(), gts[:, 3].argmin()
    lrx_full = gts[:, 0].max() + gts[xmax_i, 1] * gts[xmax_i, 2]
    lry_full = gts[:, 3].min() + gts[ymin_i, 4] * gts[ymin_i, 5]
    if len(set(gts[:, 1])) != 1 or len(set(gts[:, 4])) != 1:
        warnings.warn("Not all tiles have the same resolution.")
    MW, MH = round((lrx_full - ulx_full) / xres), round((lry_full - uly_full) / yres)
    C = model.n_out
    print(f"True merged raster size: {C * MH * MW * 4 / (1024 ** 2): .1f}MB.")
    if large_file and not regression:
        # int8 path of the reference: probabilities * 31 rounded to int8, integer division by the hit counter (host)
        merged = np.zeros((C, MH, MW), dtype=np.int8)
        counter = np.zeros((C, MH, MW), dtype=np.int8)
        for i, p in results:
            x0, y0 = round((gts[i, 0] - ulx_full) / xres), round((gts[i, 3] - uly_full) / yres)
            q = np.around(p.cpu().numpy() * ((128 / 4) - 1)).astype(np.int8)
            merged[:, y0:y0 + q.shape[1], x0:x0 + q.shape[2]] += q
            counter[:, y0:y0 + q.shape[1], x0:x0 + q.shape[2]] += 1
        if world > 1:
            raise NotImplementedError("large_file merge is single-process (int8 host arrays)")
        m = counter > 0
        merged[m] //= counter[m]
        amax_full = merged.argmax(axis=0)
    else:
        mosaic = torch.zeros((C, MH, MW), dtype=torch.float32, device=model._device)
        count = torch.zeros((MH, MW), dtype=torch.int32, device=model._device)
        for i, p in results:
            x0, y0 = round((gts[i, 0] - ulx_full) / xres), round((gts[i, 3] - uly_full) / yres)
            ops.mosaic_accumulate(p.contiguous(), mosaic, count, int(y0), int(x0))
        if world > 1:   # partial (sum-probs, hit-count) rasters of the ranks -> one mosaic
            dist.all_reduce(mosaic)
            dist.all_reduce(count)
            if rank != 0:
                return None
        am = torch.empty((MH, MW), dtype=torch.uint8, device=model._device)
        ops.mosaic_finalize(mosaic, count, am)
        merged, amax_full = mosaic.cpu().numpy(), am.cpu().numpy()
        if regression:       # predict.py:306-315: mean of the overlapping tiles, -9999 where no prediction was placed
            out = merged[0]
            out[count.cpu().numpy() == 0] = -9999
            name = "_".join(filter(None, [AOI, year, model_name, "prediction"])) + ".tif"
            store_tif(output_folder / name, out, [ulx_full, xres, 0.0, uly_full, 0.0, yres], geos[0][1], -9999, class_zero)
            print(f"Prediction stored in {output_folder}.")
            return output_folder
    if all_classes:
        out = merged
    elif specific_class is None:
        out = amax_full.astype(np.uint8)
    else:
        out = merged[specific_class]
    name = "_".join(filter(None, [AOI, year, model_name, "prediction"])) + ".tif"
    store_tif(output_folder / name, out, [ulx_full, xres, 0.0, uly_full, 0.0, yres], geos[0][1], None, class_zero)
    print(f"Prediction stored in {output_folder}.")
    return output_folder / name
