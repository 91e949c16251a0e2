"""conv1x1 + ReLU + PixelShuffle as one launch (UNET_FUSE_SHUFFLE=1, default) against conv + shuffle pass (0): cfg2 step, predict, per storage type"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from unet_amd import modules as M

dev = torch.device("cuda", 0)
for rep in range(2):
    for dtype in ("f32", "bf16"):
        for on in (False, True):
            M.FUSE_SHUFFLE = on
            r = B.step_bench("xresnet34", 4, 5, 512, 16, dtype, 8, 3, 0, 1, dev, lambda m: None, probe=False)
            p16 = B.predict_bench(dtype, 16, dev)
            p1 = B.predict_bench(dtype, 1, dev, iters=20)
            print(json.dumps({"dtype": dtype, "fused_shuffle": on, "step_tiles_per_s": round(16 * 8 / r["dt"], 2), "mem_GB": round(r["mem"] / 2**30, 2),
                              "predict_b16": p16["value"], "predict_b1": p1["value"]}), flush=True)
