import copy, sys, torch
sys.path.insert(0, ".")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
torch.set_num_threads(16)
def nchw(ts): return ts.view().permute(0, 3, 1, 2).contiguous().cpu()

def run(arch, n_in, n_out, size, B, sa, gamma=0.7):
    torch.manual_seed(21)
    ref = O.DynamicUnet(arch, n_in, n_out, size, self_attention=sa)
    O.randomize_bn_and_zero_gammas(ref, seed=22)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, O.SelfAttention): m.gamma.fill_(gamma)
    x, y = O.synthetic_batch(B, n_in, size[0], size[1], n_out)
    model = HipDynamicUnet(arch, n_in, n_out, size, self_attention=sa); model.load_state_dict(ref.state_dict())
    ref.eval(); model.eval()
    t32 = {}
    with torch.no_grad():
        z32 = ref(x, t32)
        z = model(x.cuda()).cpu()
    ctx, L = model.ctx, model.layers
    hip = {"encoder": nchw(ctx.saved[(id(model), "e")])}
    for k in range(4):
        blk = L[4 + k]; s = ctx.saved[(id(blk), "s")]
        if blk.sa is not None:
            hip[f"unet{k}"] = nchw(ctx.act(blk.sa, "out", s.N, s.H, s.W, blk.out_channels))
            hip[f"unet{k}_preSA"] = nchw(ctx.act(blk.conv2, "a", s.N, s.H, s.W, blk.out_channels))
        else:
            hip[f"unet{k}"] = nchw(ctx.act(blk.conv2, "a", s.N, s.H, s.W, blk.out_channels))
    print(f"--- {arch} {n_in}->{n_out} {size} B={B} sa={sa}: logits err {(z-z32).abs().max().item():.2e} scale {z32.abs().max().item():.2e}")
    for k in ("encoder", "unet0", "unet1", "unet2", "unet3"):
        a = t32[k]
        print(f"   tap {k:8s} rel max err {((hip[k]-a).abs().max()/a.abs().max()).item():.2e}")
    if sa:
        # oracle pre-SA activation of unet1
        blk = ref.layers[5]
        print("   (pre-SA tap available on HIP side only)")

run("xresnet34", 3, 3, (400, 400), 2, True)
run("xresnet34", 3, 3, (400, 400), 1, True)
run("xresnet34", 4, 5, (400, 400), 1, True)
run("xresnet34", 3, 3, (416, 416), 1, True)
run("xresnet34", 3, 3, (384, 384), 2, True)
run("xresnet34", 3, 3, (400, 400), 2, False)
