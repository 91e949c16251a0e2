"""End-to-end wall time of train.train_func (the reference's training entry point, train.py:287-375) on a synthetic tile dataset, next to the
kernel time of the steps it runs: what the host side adds around fit_one_cycle (class weights, model build, validation, export).
usage: python scripts/prof_train_func.py [n_train=256] [n_valid=64] [epochs=2] [f32|bf16]"""
import cProfile, json, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from unet_amd.tiffio import write_tiff

n_train, n_valid, epochs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 256), (2, 64), (3, 2)))
dt = sys.argv[4] if len(sys.argv) > 4 else "f32"
os.environ["UNET_ACT_DTYPE"] = dt
root = tempfile.mkdtemp(prefix="unet_train_")
try:
    g = np.random.default_rng(0)
    for split, n in (("trai", n_train), ("vali", n_valid)):
        for sub in ("img_tiles", "mask_tiles"):
            os.makedirs(os.path.join(root, "data", split, sub))
        for i in range(n):
            img = g.integers(1, 256, (4, 512, 512), dtype=np.uint8)
            mask = np.repeat(np.repeat(g.integers(0, 5, (16, 16), dtype=np.uint8), 32, 0), 32, 1)
            gt = (1000.0 + 300 * i, 0.5, 0.0, 5000.0, 0.0, -0.5)
            write_tiff(os.path.join(root, "data", split, "img_tiles", f"t{i:04d}.tif"), img, geotransform=gt)
            write_tiff(os.path.join(root, "data", split, "mask_tiles", f"t{i:04d}.tif"), mask, geotransform=gt)
    import train as T
    from unet_amd import xresnet34
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    learn = T.train_func(os.path.join(root, "data"), None, os.path.join(root, "models"), "run", 16, False, False, "weighted", xresnet34, epochs, 1e-4, 10, None, None,
                         "dice_multi", False, ["vali"], list("abcde"), True, None, False, None, 0.5, "", False)
    torch.cuda.synchronize()
    pr.disable()
    dt_all = time.perf_counter() - t0
    steps = epochs * (n_train // 16)
    print(json.dumps({"dtype": dt, "train_tiles": n_train, "valid_tiles": n_valid, "epochs": epochs, "wall_s": round(dt_all, 2), "steps": steps,
                      "history": open(os.path.join(root, "models", "run", "run_history.csv")).read().strip().splitlines()[-1]}))
    pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
finally:
    shutil.rmtree(root, ignore_errors=True)
