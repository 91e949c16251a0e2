"""Where save_predictions(merge=True) over tile files spends its time: cProfile of a second call + the decode pipeline alone.
usage: python scripts/prof_save_predictions.py [side=8000] [f32|bf16]"""
import cProfile, json, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import create_tiles_unet as T
import predict as P
from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, Learner, TileDataset
from unet_amd.model import HipDynamicUnet
from unet_amd.tiffio import write_tiff

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
if os.environ.get("SWITCH"):
    sys.setswitchinterval(float(os.environ["SWITCH"]))
root = tempfile.mkdtemp(prefix="unet_files_")
try:
    g = np.random.default_rng(0)
    img = g.integers(1, 256, (4, side, side), dtype=np.uint8)
    write_tiff(os.path.join(root, "scene.tif"), img, geotransform=(400000.0, 0.5, 0.0, 5700000.0, 0.0, -0.5))
    n = T.split_raster(os.path.join(root, "scene.tif"), None, os.path.join(root, "cut"), patch_size=512, patch_overlap=0.2, split=[1])["tiles"]
    torch.manual_seed(0)
    model = HipDynamicUnet("xresnet34", 4, 5, (512, 512), act_dtype=dt)
    dls = DataLoaders(TileDataset([np.zeros((4, 512, 512), np.uint8)], None, "int8"), None, 1, device="cuda", vocab=list("abcde"))
    learn = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), metrics=[DiceMulti()], path=root)
    pkl = os.path.join(root, "m.pkl")
    learn.export(pkl)
    del learn, model
    tiles_dir = os.path.join(root, "cut", "img_tiles")
    P.save_predictions(pkl, tiles_dir, False, merge=True, AOI="warm", validation_vision=False, batch_size=16)
    # the decode pipeline alone
    tiles = sorted(p for p in __import__("pathlib").Path(tiles_dir).iterdir())
    batches = [(i, min(16, len(tiles) - i)) for i in range(0, len(tiles), 16)]
    t0 = time.perf_counter()
    pf = P._TilePrefetcher(tiles, batches)
    cnt = 0
    for first, nn, buf in pf:
        d = pf.upload(buf, torch.device("cuda"))
        cnt += nn
    torch.cuda.synchronize()
    print(f"prefetcher alone: {cnt / (time.perf_counter() - t0):.0f} tiles/s")
    tm = {}
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    P.save_predictions(pkl, tiles_dir, False, merge=True, AOI="timed", validation_vision=False, batch_size=16, timing=tm)
    dtot = time.perf_counter() - t0
    pr.disable()
    print(json.dumps({"dtype": dt, "tiles": n, "total_s": round(dtot, 3), "engine_s": round(tm["seconds"], 3)}))
    if os.environ.get("STATS"):
        pstats.Stats(pr).sort_stats("cumtime").print_stats(32)
finally:
    shutil.rmtree(root, ignore_errors=True)
