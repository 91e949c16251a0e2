import sys, torch
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
from tests.util import from_ts

torch.manual_seed(0)
ref = O.DynamicUnet('xresnet34', 4, 5, (64, 64))
O.randomize_bn_and_zero_gammas(ref, seed=1)
model = HipDynamicUnet('xresnet34', 4, 5, (64, 64))
model.load_state_dict(ref.state_dict())
x, y = O.synthetic_batch(1, 4, 64, 64, 5)
ref.train(); model.train()
taps = {}
z = ref(x, taps)
for t in taps.values(): t.retain_grad()
O.CrossEntropyLossFlat()(z, y).backward()
model.forward_loss_backward(x.cuda(), y.cuda(), None)
torch.cuda.synchronize()
ctx = model.ctx
L = model.layers
for k in range(4):
    blk = L[4 + k]
    out = from_ts(ctx.act(blk.conv2, 'a', *taps[f'unet{k}'].shape[0:1], *taps[f'unet{k}'].shape[2:], blk.conv2.nf))
    r = taps[f'unet{k}'].detach()
    nxt = L[5 + k].shuf[0] if k < 3 else L[8][0]
    g = from_ts(ctx.act(nxt, 'dx', r.shape[0], r.shape[2], r.shape[3], r.shape[1]))
    gref = taps[f'unet{k}'].grad * (r > 0)
    print(f'unet{k}', tuple(r.shape), 'fwd err %.2e' % (out - r).abs().max().item(), 'mask mismatches', int(((out > 0) != (r > 0)).sum()),
          'of', r.numel(), 'grad err %.2e scale %.2e' % ((g - gref).abs().max().item(), gref.abs().max().item()),
          'frac zero %.3f' % (r == 0).float().mean().item())
    d = (g - gref).abs()
    bad = (d > 1e-3 * gref.abs().max()).nonzero()
    print('   n bad', len(bad), bad[:8].tolist())
