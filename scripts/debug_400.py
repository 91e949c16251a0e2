import sys, torch, copy
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
torch.manual_seed(0)
ref = O.DynamicUnet("xresnet34", 4, 5, (400, 400)); O.randomize_bn_and_zero_gammas(ref, seed=1)
model = HipDynamicUnet("xresnet34", 4, 5, (400, 400)); model.load_state_dict(ref.state_dict())
x, y = O.synthetic_batch(1, 4, 400, 400, 5)
ref.eval(); model.eval()
taps = {}
with torch.no_grad():
    z_ref = ref(x, taps)
    z64 = copy.deepcopy(ref).double()(x.double())
probs, amax = model.predict_probs(x.cuda())
z = model.logits_ts().view().permute(0, 3, 1, 2).cpu()
print('logit scale', z_ref.abs().max().item(), 'hip-cpu', (z - z_ref).abs().max().item(), 'cpu-f64', (z_ref.double()-z64).abs().max().item(), 'hip-f64', (z.double()-z64).abs().max().item())
ctx = model.ctx
e = ctx.saved[(id(model), 'e')]
print('encoder err', (e.view().permute(0,3,1,2).cpu() - taps['encoder']).abs().max().item(), taps['encoder'].abs().max().item())
for k in range(4):
    blk = model.layers[4+k]; r = taps[f'unet{k}']
    out = ctx.act(blk.conv2, 'a', r.shape[0], r.shape[2], r.shape[3], r.shape[1]).view().permute(0,3,1,2).cpu()
    print(f'unet{k}', tuple(r.shape), 'err', (out - r).abs().max().item(), 'scale', r.abs().max().item())
d = (z - z_ref).abs().amax(1)[0]
idx = d.argmax(); print('worst pixel', divmod(int(idx), 400), 'mismatch argmax', int((amax.cpu() != z_ref.argmax(1)).sum()))
