import sys, torch
sys.path.insert(0, ".")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
torch.set_num_threads(16)
def nchw(ts): return ts.view().permute(0, 3, 1, 2).contiguous().cpu()
torch.manual_seed(21)
size = (384, 384)
ref = O.DynamicUnet("xresnet34", 3, 3, size, self_attention=True)
O.randomize_bn_and_zero_gammas(ref, seed=22)
with torch.no_grad():
    for m in ref.modules():
        if isinstance(m, O.SelfAttention): m.gamma.fill_(0.7)
x, y = O.synthetic_batch(2, 3, size[0], size[1], 3)
model = HipDynamicUnet("xresnet34", 3, 3, size, self_attention=True); model.load_state_dict(ref.state_dict())
ref.eval(); model.eval()
t32 = {}
with torch.no_grad():
    z32 = ref(x, t32)
    zb = model(x.cuda()).clone().cpu()
    blk = model.layers[5]; sa = blk.sa
    N = 48 * 48
    Tb = nchw(model.ctx.act(sa, "T", 2, 48, 48, N)).clone()
    outb = nchw(model.ctx.act(sa, "out", 2, 48, 48, 384)).clone()
    preb = nchw(model.ctx.act(blk.conv2, "a", 2, 48, 48, 384)).clone()
    z0 = model(x[0:1].cuda()).clone().cpu()
    T0 = nchw(model.ctx.act(sa, "T", 1, 48, 48, N)).clone()
    out0 = nchw(model.ctx.act(sa, "out", 1, 48, 48, 384)).clone()
    pre0 = nchw(model.ctx.act(blk.conv2, "a", 1, 48, 48, 384)).clone()
    z1 = model(x[1:2].cuda()).clone().cpu()
    out1 = nchw(model.ctx.act(sa, "out", 1, 48, 48, 384)).clone()
print("T magnitude", Tb.abs().max().item())
for i, zi in ((0, z0), (1, z1)):
    print(f"img {i}: batch-vs-alone logits {(zb[i]-zi[0]).abs().max().item():.3e}; batch vs oracle {(zb[i]-z32[i]).abs().max().item():.3e}; alone vs oracle {(zi[0]-z32[i]).abs().max().item():.3e}; scale {z32[i].abs().max().item():.2e}")
print("pre-SA batch vs alone img0", (preb[0] - pre0[0]).abs().max().item(), "T", (Tb[0] - T0[0]).abs().max().item(), "SA out", (outb[0] - out0[0]).abs().max().item(), "img1 SA out", (outb[1] - out1[0]).abs().max().item())
u1 = t32["unet1"]
print("SA out vs oracle: batch img0", (outb[0]-u1[0]).abs().max().item(), "img1", (outb[1]-u1[1]).abs().max().item(), "alone img0", (out0[0]-u1[0]).abs().max().item(), "alone img1", (out1[0]-u1[1]).abs().max().item(), "scale", u1.abs().max().item())
