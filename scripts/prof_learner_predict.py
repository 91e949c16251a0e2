"""cProfile of Learner.predict on one 4x512x512 tile (the reference's per-tile loop, predict.py:191-193)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, Learner, TileDataset
from unet_amd.model import HipDynamicUnet
m = HipDynamicUnet("xresnet34", 4, 5, (512, 512))
dls = DataLoaders(TileDataset([np.zeros((4, 512, 512), np.uint8)], None, "int8"), None, 16, vocab=list("abcde"))
ln = Learner(dls, m, loss_func=CrossEntropyLossFlat(axis=1))
x = np.random.default_rng(0).integers(0, 255, (4, 512, 512)).astype(np.uint8)
for _ in range(3): ln.predict(x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): ln.predict(x)
torch.cuda.synchronize(); print("Learner.predict ms per tile", (time.perf_counter() - t) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): ln.predict(x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
