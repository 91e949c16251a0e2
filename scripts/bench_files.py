"""End-to-end rate of the tile-FILE prediction path (reference flow: split_raster -> tile files -> save_predictions(merge=True)):
a synthetic 4-band uint8 raster is cut into 512 x 512 GeoTIFF tiles on local disk, then predict.save_predictions reads, predicts, merges
and writes the mask.  Reports tiles/s from the first byte read to the mask on disk, next to predict_raster on the same raster.
usage: python scripts/bench_files.py [side=6000] [f32|bf16]"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
import torch
import create_tiles_unet as T
import predict as P
from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, Learner, TileDataset
from unet_amd.model import HipDynamicUnet
from unet_amd.tiffio import write_tiff, read_tiff

side = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
root = tempfile.mkdtemp(prefix="unet_files_")
try:
    g = np.random.default_rng(0)
    img = g.integers(1, 256, (4, side, side), dtype=np.uint8)
    write_tiff(os.path.join(root, "scene.tif"), img, geotransform=(400000.0, 0.5, 0.0, 5700000.0, 0.0, -0.5))
    t0 = time.perf_counter()
    n = T.split_raster(os.path.join(root, "scene.tif"), None, os.path.join(root, "cut"), patch_size=512, patch_overlap=0.2, split=[1])["tiles"]
    t_cut = time.perf_counter() - t0
    torch.manual_seed(0)
    model = HipDynamicUnet("xresnet34", 4, 5, (512, 512), act_dtype=dt)
    dls = DataLoaders(TileDataset([np.zeros((4, 512, 512), np.uint8)], None, "int8"), None, 1, device="cuda", vocab=list("abcde"))
    learn = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), metrics=[DiceMulti()], path=root)
    pkl = os.path.join(root, "m.pkl")
    learn.export(pkl)
    del learn, model
    tm = {}
    P.save_predictions(pkl, os.path.join(root, "cut", "img_tiles"), False, merge=True, AOI="warm", validation_vision=False, batch_size=16)   # warm-up (page cache, buffers)
    t0 = time.perf_counter()
    f = P.save_predictions(pkl, os.path.join(root, "cut", "img_tiles"), False, merge=True, AOI="timed", validation_vision=False, batch_size=16, timing=tm)
    t_files = time.perf_counter() - t0
    m2 = P.load_learner(pkl).model if dt == "f32" else P.load_learner(pkl, act_dtype=dt).model
    m2.eval()
    tr = {}
    P.predict_raster(m2, img[:, :1024, :2048].copy(), 512, 0.2, batch_size=16)
    t0 = time.perf_counter()
    out = P.predict_raster(m2, os.path.join(root, "scene.tif"), 512, 0.2, batch_size=16, timing=tr)
    t_raster = time.perf_counter() - t0
    same = bool(np.array_equal(read_tiff(f)[0], out))
    print(json.dumps({"what": "tile-file path vs in-memory raster path, same raster", "dtype": dt, "raster": [4, side, side], "tiles": n,
                      "split_raster_s": round(t_cut, 2), "save_predictions_s": round(t_files, 2), "files_tiles_per_s": round(n / t_files, 1),
                      "files_engine_tiles_per_s": round(n / tm["seconds"], 1), "predict_raster_s": round(t_raster, 2),
                      "raster_tiles_per_s_incl_tiff_read": round(n / t_raster, 1), "raster_engine_tiles_per_s": round(n / tr["seconds"], 1),
                      "identical_masks": same}))
finally:
    shutil.rmtree(root, ignore_errors=True)
