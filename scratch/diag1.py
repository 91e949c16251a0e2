import copy, sys, torch
sys.path.insert(0, ".")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
from unet_amd.ops import TS
torch.set_num_threads(16)

def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()

def nchw(ts): return ts.view().permute(0, 3, 1, 2).contiguous().cpu()

if "A" in sys.argv[1]:
    torch.manual_seed(5)
    ref = O.DynamicUnet("xresnet18", 3, 2, (256, 256)); O.randomize_bn_and_zero_gammas(ref, seed=6)
    model = HipDynamicUnet("xresnet18", 3, 2, (256, 256)); model.load_state_dict(ref.state_dict())
    ref64 = copy.deepcopy(ref).double()
    x, y = O.synthetic_batch(2, 3, 256, 256, 2)
    w = torch.tensor([0.5, 0.5])
    ref.train(); ref64.train(); model.train()
    t32, t64 = {}, {}
    O.CrossEntropyLossFlat(weight=w)(ref(x, t32), y).backward()
    O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double(), t64), y).backward()
    model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda()); torch.cuda.synchronize()
    ctx, L = model.ctx, model.layers
    hip = {"encoder": nchw(ctx.saved[(id(model), "e")]), "middle": nchw(ctx.act(L[3][1], "a", 2, 8, 8, 512))}
    for k in range(4):
        blk = L[4 + k]; s = ctx.saved[(id(blk), "s")]
        hip[f"unet{k}"] = nchw(ctx.act(blk.conv2, "a", s.N, s.H, s.W, blk.out_channels))
    hip["final_res"] = nchw(ctx.act(L[11], "out", 2, 256, 256, model.cat_c))
    for k in hip:
        a64 = t64[k].detach()
        print(f"tap {k:10s} max-rel err hip {((hip[k].double()-a64).abs().max()/a64.abs().max()).item():.2e}  cpu32 {((t32[k].detach().double()-a64).abs().max()/a64.abs().max()).item():.2e}   relL2 hip {rel(hip[k],a64):.2e} cpu32 {rel(t32[k].detach(),a64):.2e}")
    for (n, p), (_, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        print(f"{n:45s} e_hip {rel(p.grad.cpu(), r.grad):.2e} e_cpu {rel(q.grad, r.grad):.2e}")

if "C" in sys.argv[1]:
    from unet_amd.modules import SelfAttention, Ctx
    from unet_amd import ops
    for (H, W) in ((16, 12), (50, 50), (10, 10), (20, 20)):
        torch.manual_seed(1)
        C = 64
        sa_ref = O.SelfAttention(C); sa_ref.eval()
        with torch.no_grad(): sa_ref.gamma.fill_(0.7)
        sa = SelfAttention(C).cuda(); sa.load_state_dict(sa_ref.state_dict()); sa.eval()
        x = torch.randn(2, C, H, W)
        with torch.no_grad(): yr = sa_ref(x)
        ctx = Ctx(torch.device("cuda")); ctx.training = False
        xt = TS(x.permute(0, 2, 3, 1).contiguous().cuda(), 0, C)
        out = sa.hip_fwd(ctx, xt); torch.cuda.synchronize()
        print("SA", H, W, "N", H * W, "max err", (nchw(out) - yr).abs().max().item(), "scale", yr.abs().max().item())
        # stage checks
        N = H * W; c8 = C // 8; CQ = 2 * c8 + C
        qkv = nchw(ctx.act(sa, "qkv", 2, H, W, CQ)).reshape(2, CQ, N)
        f, g, h = qkv[:, :c8], qkv[:, c8:2 * c8], qkv[:, 2 * c8:]
        S = torch.bmm(f.transpose(1, 2), g)               # [B, i, j]
        T = nchw(ctx.act(sa, "T", 2, H, W, N)).reshape(2, N, N)   # [B, i(channel), j(pixel)]
        print("   T err", (T - S).abs().max().item())
        P = nchw(ctx.act(sa, "P", 2, H, W, N)).reshape(2, N, N)
        beta = torch.softmax(S, dim=1)
        print("   P err", (P - beta).abs().max().item())
        Oh = nchw(ctx.act(sa, "O", 2, H, W, C)).reshape(2, C, N)
        print("   O err", (Oh - torch.bmm(h, beta)).abs().max().item())
