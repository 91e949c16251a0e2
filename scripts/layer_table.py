"""Per-launch table of the dominant conv kernel inside one training step (cfg2): shape, ms, algorithmic TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep

torch.manual_seed(0)
DT = os.environ.get("UNET_DTYPE", "f32")       # f32 | bf16
m = HipDynamicUnet("xresnet34", 4, 5, (512, 512), act_dtype=DT); m.train()
st = TrainStep(m, FlatAdam(m, [1e-5, 3e-5, 1e-4]), torch.full((5,), 0.2, device="cuda"))
g = torch.Generator().manual_seed(1)
x = (torch.randint(0, 256, (16, 4, 512, 512), generator=g).float() / 255).cuda(); y = torch.randint(0, 5, (16, 512, 512), generator=g).cuda()
for _ in range(3): st(x, y)
torch.cuda.synchronize()
ops.CONV_PROBE = pr = ops.ConvProbe(int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 10000 + 128 * 10)
st(x, y)
torch.cuda.synchronize()
ops.CONV_PROBE = None
rows = [(a.elapsed_time(b), d) for (a, b), d in zip(pr.events, pr.detail)]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} launches, {tot:.2f} ms, {pr.flops / tot / 1e9:.1f} TFLOP/s")
for ms, (what, N, H, W, Ci, Co, ks, s_, fl) in sorted(rows, key=lambda r: -r[0]):
    print(f"{ms:7.3f} ms {fl / ms / 1e9:6.1f} TF  {what:13s} {N}x{H}x{W} {Ci:4d}->{Co:4d} k{ks} s{s_}")
