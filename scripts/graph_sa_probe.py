"""does a hipGraph capture of a training step with SelfAttention survive?  python scripts/graph_sa_probe.py <fused 0|1> <dtype>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
from unet_amd.modules import SelfAttention
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep
SelfAttention.fused = bool(int(sys.argv[1]))
dt = sys.argv[2]
size = (128, 160)
m = HipDynamicUnet("xresnet34", 3, 4, size, self_attention=True, act_dtype=dt)
m.train()
step = TrainStep(m, FlatAdam(m, [1e-4, 3e-4, 1e-3]), None, 1, use_graph=True)
for s in range(4):
    x, y = O.synthetic_batch(2, 3, size[0], size[1], 4, seed=s)
    l = step(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    print("step", s, float(l), flush=True)
print("ok", sys.argv[1:], flush=True)
