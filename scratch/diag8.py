import sys, torch, copy
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
torch.set_num_threads(16)
arch, n_in, n_out, size, bs = "xresnet34", 4, 5, (64, 64), 1
torch.manual_seed(0)
ref = O.DynamicUnet(arch, n_in, n_out, size); O.randomize_bn_and_zero_gammas(ref, seed=1)
model = HipDynamicUnet(arch, n_in, n_out, size); model.load_state_dict(ref.state_dict())
x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
w = torch.rand(n_out) + 0.5
ref.train(); model.train()
O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
for rep in range(2):
    model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda()); torch.cuda.synchronize()
    bad = []
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        e = ((p.grad.cpu().double() - q.grad.double()).norm() / (q.grad.double().norm() + 1e-30)).item()
        if e > 0.05: bad.append((n, round(e, 3)))
    print("rep", rep, len(bad), "bad tensors; last (closest to the loss) ones:", bad[-6:])
# ---- post-encoder BN in detail
ref64 = copy.deepcopy(ref).double(); ref64.train()
t64 = {}
ref64(x.double(), t64)
e64 = t64["encoder"].detach()
ctx = model.ctx
def nchw(ts): return ts.view().permute(0, 3, 1, 2).contiguous().cpu()
e_h = nchw(ctx.saved[(id(model), "e")]).double()
print("encoder out err", (e_h - e64).abs().max().item(), "scale", e64.abs().max().item())
bn = ref64.layers[1]
mean, var = e64.mean((0, 2, 3)), e64.var((0, 2, 3), unbiased=False)
pre = (e64 - mean[None, :, None, None]) / (var[None, :, None, None] + 1e-5).sqrt() * bn.weight[None, :, None, None] + bn.bias[None, :, None, None]
m0_h = nchw(ctx.act(model, "m0", 1, 2, 2, 512)).double()
print("m0 err", (m0_h - pre.clamp(min=0)).abs().max().item(), "mask mismatches", int(((m0_h > 0) != (pre > 0)).sum()), "min |pre|", pre.abs().min().item())
bx = model._post_bx
print("mean err", (ctx.vec(bx, "mean", 512).cpu().double() - mean).abs().max().item(), "invstd rel err", ((ctx.vec(bx, "invstd", 512).cpu().double() * (var + 1e-5).sqrt()) - 1).abs().max().item())
# gradient wrt m0 from the oracle: hook
ref64.zero_grad()
