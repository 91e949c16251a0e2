"""A/B inside one process: the training feed with CPython's default GIL switch interval (5 ms) against 0.5 / 0.1 ms while the feeder
runs (UNET_FEED_SWITCH_S, opt-in); fit_one_cycle over uncompressed / LZW / JPEG tile files, fp32 and bf16 storage (bench.fit_files_bench)."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {}
for name, val in (("switch_5ms", "1.0"), ("switch_0.5ms", "0.0005"), ("switch_0.1ms", "0.0001")):
    os.environ["UNET_FEED_SWITCH_S"] = val          # >= the interpreter's 5 ms: left alone
    r = bench.fit_files_bench(dev, lambda m: print(m, file=sys.stderr, flush=True), {"f32": None, "bf16": None}, n_tiles=512)
    out[name] = {k: v["value"] for k, v in r.items() if isinstance(v, dict)}
    print(name, json.dumps(out[name]), flush=True)
