import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
from unet_amd import ops
from tests.util import to_ts, empty_ts, from_ts

torch.manual_seed(0)
ref = O.DynamicUnet('xresnet34', 4, 5, (64, 64))
O.randomize_bn_and_zero_gammas(ref, seed=1)
model = HipDynamicUnet('xresnet34', 4, 5, (64, 64))
model.load_state_dict(ref.state_dict())
x, y = O.synthetic_batch(1, 4, 64, 64, 5)
model.train()
model.forward_loss_backward(x.cuda(), y.cuda(), None)
torch.cuda.synchronize()
ctx = model.ctx
blk2 = model.layers[6]
cl = blk2.shuf[0]
up_in = ctx.saved[(id(cl), 'x')]
dyc = ctx.act(cl, 'dy', up_in.N, up_in.H, up_in.W, 4 * blk2.shuf.nf)
dx = ctx.act(cl, 'dx', up_in.N, up_in.H, up_in.W, up_in.C)
w = cl[0].weight.detach().cpu()
dyc_c = from_ts(dyc); up_c = from_ts(up_in)
exp = torch.nn.grad.conv2d_input(up_c.shape, w, dyc_c) * (up_c > 0)
got = from_ts(dx)
print('shapes', up_c.shape, dyc_c.shape, 'err', (got - exp).abs().max().item(), 'scale', exp.abs().max().item())
d = (got - exp).abs()
bad = (d > 1e-3 * exp.abs().max()).nonzero()
print('n bad', len(bad), bad[:20].tolist())
# same op standalone
dxt = empty_ts(1, 8, 8, up_in.C)
ops.conv2d_dgrad(to_ts(dyc_c), ops.pack_weights(w.cuda(), 1), dxt, 1, 1, mask=to_ts(up_c))
torch.cuda.synchronize()
print('standalone err', (from_ts(dxt) - exp).abs().max().item())
dxt2 = empty_ts(1, 8, 8, up_in.C)
ops.conv2d_dgrad(to_ts(dyc_c), ops.pack_weights(w.cuda(), 1), dxt2, 1, 1)
torch.cuda.synchronize()
print('standalone nomask err', (from_ts(dxt2) - torch.nn.grad.conv2d_input(up_c.shape, w, dyc_c)).abs().max().item())
# block1 conv2 wgrad / bias check
blk1 = model.layers[5]
t1 = ctx.saved[(id(blk1.conv2), 'x')]
gw = torch.nn.grad.conv2d_weight(from_ts(t1), blk1.conv2[0].weight.shape, got, padding=1)
print('conv2 wgrad consistency', (blk1.conv2[0].weight.grad.cpu() - gw).abs().max().item(), gw.abs().max().item())
print('conv2 bias consistency', (blk1.conv2[0].bias.grad.cpu() - got.sum((0, 2, 3))).abs().max().item())
