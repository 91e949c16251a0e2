"""Weight gradients on the side stream: which launches should leave the main stream?  One process, the threshold (GFLOP of the launch) swept
per storage type: python scripts/ab_side.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from unet_amd import modules as M

dev = torch.device("cuda", 0)
log = lambda m: None
orig = M.Ctx.__init__


def run(what, thr, **kw):
    def init(self, device, act_dtype=torch.float32):
        orig(self, device, act_dtype)
        self.wgrad_overlap = thr > 0
        self.wgrad_overlap_pixels = (1 << 62) if thr > 0 else 0
        self.wgrad_overlap_min_pixels = 0
    M.Ctx.__init__ = init
    r = B.step_bench(*kw["args"], dev, log, probe=False)
    M.Ctx.__init__ = orig
    n = kw["args"][4] * kw["args"][6]
    print(json.dumps({"what": what, "threshold_gflop": thr, "tiles_per_s": round(n / r["dt"], 2), "ms_per_step": round(r["dt"] / kw["args"][6] * 1e3, 3)}), flush=True)


for thr in (0, 5, 20, 60, 200, 1e9):
    run("cfg2 f32", thr, args=("xresnet34", 4, 5, 512, 16, "f32", 8, 3, 0, 1))
for thr in (0, 20, 100, 400, 1e9):
    run("cfg2 bf16", thr, args=("xresnet34", 4, 5, 512, 16, "bf16", 10, 3, 0, 1))
for thr in (0, 1, 5, 1e9):
    run("cfg1 f32", thr, args=("xresnet18", 3, 2, 256, 2, "f32", 30, 5, 0, 1))
