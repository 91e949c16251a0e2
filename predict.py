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
from unet_amd.tiffio import read_tiff, tiff_info, write_tiff


def store_tif(output_file, data, geotrans=None, tags=None, nodata=None, class_zero=False):
    """predict.py:19-52: GeoTIFF writer; class_zero shifts class ids back by one (0 was reserved for nodata)."""
    a = np.asarray(data)
    if class_zero and a.dtype.kind in "ui":
        a = a - 1 if a.dtype.kind == "i" else (a.astype(np.int16) - 1)
    write_tiff(output_file, a, geotransform=geotrans, tags=tags, nodata=nodata)


def _geo(path):
    """(geotransform, GeoTIFF tags, height, width) from the header alone (.npy tiles carry no georeference)"""
    if Path(path).suffix == ".npy":
        a = np.load(path, mmap_mode="r")
        return None, {}, int(a.shape[-2]), int(a.shape[-1])
    meta = tiff_info(path)
    return meta["geotransform"], meta["tags"], meta["height"], meta["width"]


LARGE_FILE_SCALE = (128 / 4) - 1        # predict.py:209-214: probabilities stretched to int8 as around(p * 31)


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
    dev = model._device
    C = model.n_out
    int8_merge = bool(merge and large_file and not regression)
    mosaic = count = None
    if merge:
        # overlap merge (predict.py:257-355).  The extent follows from the tiles' geotransforms and sizes, which are known from the
        # headers BEFORE any tile is predicted: every batch is accumulated into the device mosaic as soon as it is computed and its
        # probabilities are dropped (the reference keeps all tiles' probabilities until the end).
        gts = np.array([[g[0][0], g[3], g[0][1], g[0][3], g[2], g[0][5]] for g in geos], dtype=np.float64)
        ulx_full, uly_full = gts[:, 0].min(), gts[:, 3].max()
        xres, yres = gts[0, 2], gts[0, 5]
        xmax_i, ymin_i = gts[:, 0].argmax(), gts[:, 3].argmin()
        lrx_full = gts[:, 0].max() + gts[xmax_i, 1] * gts[xmax_i, 2]
        lry_full = gts[:, 3].min() + gts[ymin_i, 4] * gts[ymin_i, 5]
        if len(set(gts[:, 1])) != 1 or len(set(gts[:, 4])) != 1:
            warnings.warn("Not all tiles have the same resolution.")
        MW, MH = round((lrx_full - ulx_full) / xres), round((lry_full - uly_full) / yres)
        print(f"True merged raster size: {C * MH * MW * (1 if int8_merge else 4) / (1024 ** 2): .1f}MB.")
        if int8_merge:      # the reference's int8 arrays (merged raster AND hit counter are int8 per class, wrap-around included)
            mosaic = torch.zeros((C, MH, MW), dtype=torch.int8, device=dev)
            count = torch.zeros((C, MH, MW), dtype=torch.int8, device=dev)
        else:
            mosaic = torch.zeros((C, MH, MW), dtype=torch.float32, device=dev)
            count = torch.zeros((MH, MW), dtype=torch.int32, device=dev)
    mine = list(range(rank, len(tiles), world))
    for b0 in range(0, len(mine), batch_size):
        ids = mine[b0:b0 + batch_size]
        chunk = [tiles[i] for i in ids]
        x = torch.from_numpy(np.stack([scale_input(open_tile(t), dtype) for t in chunk]))
        if regression:       # predict.py:195-197: tile_preds[1] = raw outputs [1,H,W]
            probs, amax = model.predict_values(x.to(dev)), None
        else:
            probs, amax = model.predict_probs(x.to(dev))
        for j, t in enumerate(chunk):
            i = ids[j]
            gt, tags = geos[i][0], geos[i][1]
            if merge:
                x0, y0 = round((gts[i, 0] - ulx_full) / xres), round((gts[i, 3] - uly_full) / yres)
                if int8_merge:
                    q = torch.round(probs[j] * LARGE_FILE_SCALE).to(torch.int8)          # np.around: half to even, as torch.round
                    mosaic[:, y0:y0 + q.shape[1], x0:x0 + q.shape[2]] += q
                    count[:, y0:y0 + q.shape[1], x0:x0 + q.shape[2]] += 1
                else:
                    ops.mosaic_accumulate(probs[j].contiguous(), mosaic, count, int(y0), int(x0))
                continue
            if regression or all_classes:
                out = probs[j].cpu().numpy()
            elif specific_class is None:
                out = amax[j].cpu().numpy().astype(np.uint8)
            else:
                out = probs[j, specific_class].cpu().numpy()
            if large_file and out.dtype.kind == "f" and out.max() <= 1 and (all_classes or specific_class):
                out = np.around(out * LARGE_FILE_SCALE).astype(np.int8)
            name = t.name if t.suffix != ".npy" else t.stem + ".tif"
            store_tif(output_folder / name, out, gt, tags, None, class_zero)
    if validation_vision:
        pass  # per-tile majority-class confusion plots (predict.py:56-143) are reporting, out of scope
    if not merge:
        if world > 1:
            dist.barrier()
        print(f"Prediction stored in {output_folder}.")
        return output_folder
    if world > 1:       # partial rasters of the ranks -> one mosaic on every rank; rank 0 writes
        if int8_merge:
            # int8 + int8 wraps modulo 256 in the single-process reference; a SUM in int32 followed by the wrapping cast back to int8
            # is the same number whatever the order of the tiles
            for t8 in (mosaic, count):
                t32 = t8.to(torch.int32)
                dist.all_reduce(t32)
                t8.copy_(t32.to(torch.int8))
        else:
            dist.all_reduce(mosaic)
            dist.all_reduce(count)
        if rank != 0:
            return None
    if int8_merge:
        merged, counter = mosaic.cpu().numpy(), count.cpu().numpy()
        m = counter > 0
        merged[m] //= counter[m]                       # predict.py:324-329: integer floor division, numpy semantics
        amax_full = merged.argmax(axis=0)
    else:
        am = torch.empty((MH, MW), dtype=torch.uint8, device=dev)
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
