"""cfg1 training step / predict at batch 1 for rocprofv3 --kernel-trace --stats (what the small configurations spend their time on)"""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep
which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
torch.manual_seed(0)
if which == "cfg1":
    m = HipDynamicUnet("xresnet18", 3, 2, (256, 256)); m.train()
    opt = FlatAdam(m, [1e-5, 3e-5, 1e-4]); st = TrainStep(m, opt, torch.full((2,), 0.5, device="cuda"))
    g = torch.Generator().manual_seed(1)
    x = (torch.randint(0, 256, (2, 3, 256, 256), generator=g).float() / 255).cuda(); y = torch.randint(0, 2, (2, 256, 256), generator=g).cuda()
    for _ in range(20): st(x, y)
else:
    m = HipDynamicUnet("xresnet34", 4, 5, (512, 512), act_dtype=os.environ.get("UNET_DTYPE", "f32")); m.eval()
    x = torch.rand(16 if which == "b16" else 1, 4, 512, 512, device="cuda")
    for _ in range(20): m.predict_probs(x)
torch.cuda.synchronize()
