"""Where does the HOST spend a cfg1 step?  cProfile over 60 eager steps (xresnet18, 3x256^2, batch 2): python scripts/host_profile.py [f32|bf16]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
torch.manual_seed(0)
m = HipDynamicUnet("xresnet18", 3, 2, (256, 256), act_dtype=dt); m.train()
opt = FlatAdam(m, [1e-5, 3e-5, 1e-4]); st = TrainStep(m, opt, torch.full((2,), 0.5, device="cuda"))
g = torch.Generator().manual_seed(1)
x = (torch.randint(0, 256, (2, 3, 256, 256), generator=g).float() / 255).cuda(); y = torch.randint(0, 2, (2, 256, 256), generator=g).cuda()
for _ in range(10): st(x, y)
torch.cuda.synchronize()
# host-only time: issue 60 steps without waiting for the GPU, stop the clock before the sync
t0 = time.perf_counter()
for _ in range(60): st(x, y)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"[{dt}] host issue time {t_host / 60 * 1e3:.3f} ms / step, wall {t_all / 60 * 1e3:.3f} ms / step")
pr = cProfile.Profile()
pr.enable()
for _ in range(60): st(x, y)
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime"); ps.print_stats(28)
