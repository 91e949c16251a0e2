"""A/B of the two MFMA shapes on the whole training step (one process, interleaved rounds)."""
import sys, time, torch
sys.path.insert(0, '.')
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
from unet_amd._lib import lib
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep
torch.manual_seed(0)
model = HipDynamicUnet('xresnet34', 4, 5, (512, 512)); model.train()
opt = FlatAdam(model, [1e-5, 3e-5, 1e-4]); step = TrainStep(model, opt, torch.full((5,), 0.2, device='cuda'))
g = torch.Generator().manual_seed(1)
x = (torch.randint(0, 256, (16, 4, 512, 512), generator=g).float() / 255).cuda(); y = torch.randint(0, 5, (16, 512, 512), generator=g).cuda()
for _ in range(2): step(x, y)
torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(4):
    for shape in (0, 1):
        _knobs.set_knob("wgrad_narrow", shape)
        step(x, y); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): step(x, y)
        torch.cuda.synchronize()
        res[shape].append((time.perf_counter() - t0) / 3 * 1e3)
for k, v in res.items():
    print(f"wgrad narrow={k}: ms/step median {sorted(v)[len(v)//2]:.2f} min {min(v):.2f}  -> {16e3/min(v):.1f} tiles/s", flush=True)
