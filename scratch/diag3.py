import sys, torch
sys.path.insert(0, ".")
from oracle import unet_oracle as O
from unet_amd.modules import SelfAttention, Ctx
from unet_amd.ops import TS
torch.set_num_threads(16)
def nchw(ts): return ts.view().permute(0, 3, 1, 2).contiguous().cpu()
for (B, C, H, W) in ((2, 384, 48, 48), (1, 384, 48, 48), (2, 384, 16, 16), (2, 128, 48, 48), (3, 256, 32, 32), (2, 384, 32, 32)):
    torch.manual_seed(1)
    sa_ref = O.SelfAttention(C); sa_ref.eval()
    with torch.no_grad(): sa_ref.gamma.fill_(0.7)
    sa = SelfAttention(C).cuda(); sa.load_state_dict(sa_ref.state_dict()); sa.eval()
    x = torch.randn(B, C, H, W)
    with torch.no_grad(): yr = sa_ref(x)
    ctx = Ctx(torch.device("cuda")); ctx.training = False
    xt = TS(x.permute(0, 2, 3, 1).contiguous().cuda(), 0, C)
    out = sa.hip_fwd(ctx, xt); torch.cuda.synchronize()
    print("SA B", B, "C", C, H, W, "N", H * W, "max err", (nchw(out) - yr).abs().max().item(), "scale", yr.abs().max().item())
    N = H * W; c8 = C // 8; CQ = 2 * c8 + C
    qkv = nchw(ctx.act(sa, "qkv", B, H, W, CQ)).reshape(B, CQ, N)
    f, g, h = qkv[:, :c8], qkv[:, c8:2 * c8], qkv[:, 2 * c8:]
    S = torch.bmm(f.transpose(1, 2), g)
    T = nchw(ctx.act(sa, "T", B, H, W, N)).reshape(B, N, N)
    for b in range(B): print(f"   img {b}: T err {(T[b] - S[b]).abs().max().item():.2e}", end="")
    P = nchw(ctx.act(sa, "P", B, H, W, N)).reshape(B, N, N)
    beta = torch.softmax(S, dim=1)
    Oh = nchw(ctx.act(sa, "O", B, H, W, C)).reshape(B, C, N)
    Or = torch.bmm(h, beta)
    for b in range(B): print(f"   img {b}: P err {(P[b] - beta[b]).abs().max().item():.2e} O err {(Oh[b] - Or[b]).abs().max().item():.2e}", end="")
    print()
