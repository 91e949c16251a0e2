import sys, torch
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
torch.manual_seed(5)
size=(64,64)
ref = O.DynamicUnet("xresnet34", 4, 5, size, self_attention=True)
O.randomize_bn_and_zero_gammas(ref, seed=6)
sa = ref.layers[5].conv2[2]
for gam in (0.0, 0.6):
    with torch.no_grad(): sa.gamma.fill_(gam)
    model = HipDynamicUnet("xresnet34", 4, 5, size, self_attention=True)
    model.load_state_dict(ref.state_dict())
    x, y = O.synthetic_batch(2, 4, 64, 64, 5)
    ref.eval(); model.eval()
    taps={}
    with torch.no_grad():
        z_ref = ref(x, taps); z = model(x.cuda()).cpu()
    print('gamma', gam, 'logit err', (z-z_ref).abs().max().item(), 'logit scale', z_ref.abs().max().item())
    ctx=model.ctx; blk=model.layers[5]
    out = ctx.act(blk.sa,'out',2,8,8,384).view().permute(0,3,1,2).cpu()
    print('  unet1 err', (out-taps['unet1']).abs().max().item(), 'scale', taps['unet1'].abs().max().item())
    ref64 = __import__('copy').deepcopy(ref).double().eval()
    with torch.no_grad(): z64 = ref64(x.double())
    print('  cpu32 vs f64', (z_ref.double()-z64).abs().max().item(), 'hip vs f64', (z.double()-z64).abs().max().item())
