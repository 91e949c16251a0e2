"""A/B in one process: cfg2 step (bf16 storage) without SelfAttention, with the blockwise products, with the fused kernels (csrc/attention.hip);
then predict at batch 16.  python scripts/ab_sa_fused.py [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from unet_amd.modules import SelfAttention

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
quiet = lambda m: None


def run(dt, sa, fused):
    SelfAttention.fused = fused
    r = B.step_bench(os.environ.get("UNET_AB_ARCH", "xresnet34"), 4, 5, 512, int(os.environ.get("UNET_AB_BATCH", "16")), dt, steps, 3, 0, 1, dev, quiet,
                     probe=False, self_attention=sa)
    torch.cuda.empty_cache()
    return int(os.environ.get("UNET_AB_BATCH", "16")) * steps / r["dt"], r["dt"] / steps * 1e3, r["loss"]


for dt in (sys.argv[2:] or ["bf16"]):
    for rep in range(2):
        for name, sa, fused in (("SA off", False, True), ("SA blockwise", True, False), ("SA fused", True, True)):
            v, ms, loss = run(dt, sa, fused)
            print(f"{dt} {name:13s} rep {rep}: {v:8.2f} tiles/s  {ms:7.3f} ms/step  loss {loss:.5f}", flush=True)
