"""Which Python call sites issue ATen ops (fill / copy / ... kernels that show up as FillFunctor / copyBuffer launches in the kernel trace)
inside one training step: python scripts/find_fills.py [cfg1|cfg2] [f32|bf16] [sa]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep

which, dt = (sys.argv + ["cfg2", "bf16"])[1:3]
sa = "sa" in sys.argv
torch.manual_seed(0)
if which == "cfg1":
    arch, n_in, n_cls, size, B = "xresnet18", 3, 2, 256, 2
else:
    arch, n_in, n_cls, size, B = "xresnet34", 4, 5, 512, 16
m = HipDynamicUnet(arch, n_in, n_cls, (size, size), act_dtype=dt, self_attention=sa); m.train()
opt = FlatAdam(m, [1e-5, 3e-5, 1e-4]); st = TrainStep(m, opt, torch.full((n_cls,), 1.0 / n_cls, device="cuda"))
g = torch.Generator().manual_seed(1)
x = (torch.randint(0, 256, (B, n_in, size, size), generator=g).float() / 255).cuda(); y = torch.randint(0, n_cls, (B, size, size), generator=g).cuda()
for _ in range(3): st(x, y)
torch.cuda.synchronize()
counts = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        fr = [s for s in traceback.extract_stack()[:-1] if "unet_amd" in s.filename][-2:]
        counts[(str(func), " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


with Log():
    st(x, y)
torch.cuda.synchronize()
for (n, site), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{c:5d}  {n:34s} {site}")
