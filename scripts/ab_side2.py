"""side-stream weight gradients x the bench's conv probe (timing events around the large-layer launches): python scripts/ab_side2.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from unet_amd import modules as M

dev = torch.device("cuda", 0)
orig = M.Ctx.__init__
for rep in range(2):
    for dtype in ("f32", "bf16"):
        for thr in (0, 1e9):
            for probe in (False, True):
                def init(self, device, act_dtype=torch.float32, _t=thr):
                    orig(self, device, act_dtype)
                    self.wgrad_overlap = _t > 0
                    self.wgrad_overlap_pixels = (1 << 62) if _t > 0 else 0
                    self.wgrad_overlap_min_pixels = 0
                M.Ctx.__init__ = init
                r = B.step_bench("xresnet34", 4, 5, 512, 16, dtype, 8, 3, 0, 1, dev, lambda m: None, probe=probe)
                M.Ctx.__init__ = orig
                ps = r["probe"]
                print(json.dumps({"dtype": dtype, "side": thr > 0, "probe": probe, "tiles_per_s": round(16 * 8 / r["dt"], 2),
                                  "probe_avg_ms": None if ps is None else round(ps["avg_ms"], 4)}), flush=True)
