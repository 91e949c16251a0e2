"""Fused SelfAttention kernels for bf16 storage (csrc/attention.hip) against plain torch on the same bf16-representable operands.

fastai layers.py SelfAttention (reference params_and_main.py:81-83: the shipped default): beta = softmax(f^T g, dim=1), o = h beta.
Tolerances (bf16 storage, stated here): the fused kernels round the un-normalised weights exp(t - max) and the softmax adjoint dT to bf16 before
their second product and the outputs once more, so every output tensor is compared in relative L2 against an fp64 evaluation of the same
formulas: O <= 6e-3, lse <= 5e-5 x max(1, |lse|) (fp32 arithmetic with v_exp_f32 / v_log_f32), dF / dG / dH <= 1.2e-2.  The module-level tests compare the fused
path with the blockwise path of the same library (unet_amd.modules.SelfAttention.fused = False), which has the same rounding points.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _qkv(B, H, W, dp, d, C, seed, scale=1.0):
    """random fused QKV buffer [B][H][W][2 dp + C] (bf16) with zero pad lanes d .. dp in the query / key slices"""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, W, 2 * dp + C, generator=g) * scale
    q[..., 2 * dp:] *= 1.0 / scale
    q[..., d:dp] = 0
    q[..., dp + d:2 * dp] = 0
    return q.to(torch.bfloat16).cuda().contiguous()


def _ref(qkv, dp, C, dO=None):
    """fp64 on the device: O, lse and (with dO) the gradient of the QKV buffer"""
    B, H, W, CQ = qkv.shape
    N = H * W
    x = qkv.double().view(B, N, CQ).requires_grad_(dO is not None)
    F_, G_, H_ = x[..., :dp], x[..., dp:2 * dp], x[..., 2 * dp:2 * dp + C]
    T = G_ @ F_.transpose(1, 2)                      # [B][j][i]
    lse = torch.logsumexp(T, dim=2)
    O = torch.softmax(T, dim=2) @ H_
    if dO is None:
        return O, lse, None
    O.backward(dO.double().view(B, N, C))
    return O.detach(), lse.detach(), x.grad


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


SHAPES = [
    # B, H,  W,  d,  dp, C
    (2, 12, 16, 48, 48, 384),       # the shipped width, three key blocks
    (8, 16, 16, 48, 48, 384),       # batch a multiple of 8: the one-image-per-XCD workgroup order
    (1, 10, 13, 48, 48, 384),       # 130 positions: ragged key block and ragged query tiles
    (3, 25, 25, 54, 56, 432),       # xresnet34_deep: 54 -> 56 query lanes, 27 channel tiles (the 32-tile instantiation), 625 positions
    (1, 8, 8, 16, 16, 128),         # narrow: one channel-tile group, unused steps of the reduction guarded off
    (1, 96, 96, 48, 48, 384),       # 9216 positions (a 768 x 768 tile): 144 key blocks, nothing of size N x N anywhere but in this test's reference
]


def test_pack_image_layout():
    from unet_amd import ops
    from unet_amd.ops import TS
    B, H, W, cs, co, cc = 2, 9, 11, 64, 8, 40
    N = H * W
    x = torch.randn(B, H, W, cs).to(torch.bfloat16).cuda()
    out = torch.full((B * ops.sa_pack_elems(N, cc),), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.sa_pack(TS(x, co, cc), out)
    torch.cuda.synchronize()
    NKB, NT = (N + 63) // 64, (cc + 15) // 16
    img = out.view(B, NKB, NT, 2, 64, 8).float().cpu()
    xs = torch.zeros(B, NKB * 64, NT * 16)
    xs[:, :N, :cc] = x.float().cpu().view(B, N, cs)[..., co:co + cc]
    lane = torch.arange(64)
    for blk in range(NKB):
        for ct in range(NT):
            for ks in range(2):
                rows = blk * 64 + ks * 32 + (lane // 16)[:, None] * 8 + torch.arange(8)[None, :]          # [lane][e]
                cols = (ct * 16 + lane % 16)[:, None].expand(64, 8)
                assert torch.equal(img[:, blk, ct, ks], xs[:, rows, cols]), (blk, ct, ks)


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_forward_against_fp64(shape):
    from unet_amd import ops
    from unet_amd.ops import TS
    B, H, W, d, dp, C = shape
    N = H * W
    qkv = _qkv(B, H, W, dp, d, C, seed=5, scale=0.6)
    O_ref, lse_ref, _ = _ref(qkv, dp, C)
    O = torch.full((B, H, W, C + 8), 9.0, dtype=torch.bfloat16, device="cuda")
    Np = ops.sa_rows(N)
    lse = torch.zeros(B * Np, device="cuda")
    pk = torch.zeros(B * ops.sa_pack_elems(N, C), dtype=torch.bfloat16, device="cuda")
    q = TS(qkv, 0, 2 * dp + C)
    ops.sa_pack(q.sub(2 * dp, C), pk)
    ops.sa_fwd(q, dp, C, pk, TS(O, 8, C), lse)
    torch.cuda.synchronize()
    assert torch.all(O[..., :8] == 9.0), "channels outside the slice were written"
    rO = _rel(O[..., 8:].view(B, N, C), O_ref)
    el = (lse.view(B, Np)[:, :N].double() - lse_ref).abs().max().item()
    assert torch.all(lse.view(B, Np)[:, N:] == 1e30), "lse of the rows past N"
    print(f"sa_fwd {shape}: O rel-L2 {rO:.3e}  lse max abs err {el:.3e} (max |lse| {lse_ref.abs().max().item():.2f})")
    assert rO <= 6e-3
    assert el <= 5e-5 * max(1.0, lse_ref.abs().max().item())


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_backward_against_fp64(shape):
    from unet_amd import ops
    from unet_amd.ops import TS
    B, H, W, d, dp, C = shape
    N, CQ = H * W, 2 * dp + C
    qkv = _qkv(B, H, W, dp, d, C, seed=6, scale=0.6)
    dO = (torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(7))).to(torch.bfloat16).cuda()
    O_ref, lse_ref, g_ref = _ref(qkv, dp, C, dO)
    q = TS(qkv, 0, CQ)
    O = O_ref.to(torch.bfloat16).view(B, H, W, C).contiguous()
    Np = ops.sa_rows(N)
    lse = torch.full((B, Np), 1e30, device="cuda")
    lse[:, :N] = lse_ref.float()
    D = torch.zeros(B * Np, device="cuda")
    ops.sa_rowdot(TS(dO, 0, C), TS(O, 0, C), D)
    D_ref = (dO.double().view(B, N, C) * O.double().view(B, N, C)).sum(-1)
    pk, gpk, fpk = (torch.zeros(B * ops.sa_pack_elems(N, c), dtype=torch.bfloat16, device="cuda") for c in (C, dp, dp))
    ops.sa_pack(TS(dO, 0, C), pk)
    ops.sa_pack(q.sub(dp, dp), gpk)
    ops.sa_pack(q.sub(0, dp), fpk)
    dqkv = torch.full((B, H, W, CQ), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.sa_bwd(q, dp, C, TS(dO, 0, C), pk, gpk, fpk, lse, D, TS(dqkv, 0, CQ))
    torch.cuda.synchronize()
    assert (D.view(B, Np)[:, :N].double() - D_ref).abs().max().item() <= 1e-4 * D_ref.abs().max().item() + 1e-5
    g = dqkv.view(B, N, CQ)
    rF, rG, rH = _rel(g[..., :d], g_ref[..., :d]), _rel(g[..., dp:dp + d], g_ref[..., dp:dp + d]), _rel(g[..., 2 * dp:], g_ref[..., 2 * dp:2 * dp + C])
    print(f"sa_bwd {shape}: rel-L2 dF {rF:.3e} dG {rG:.3e} dH {rH:.3e}")
    assert rF <= 1.2e-2 and rG <= 1.2e-2 and rH <= 1.2e-2
    if dp > d:
        assert torch.all(g[..., d:dp] == 0) and torch.all(g[..., dp + d:2 * dp] == 0), "pad lanes of dF / dG must be exact zeros"
    # no atomics anywhere: a second run gives the same bits
    dq2 = torch.zeros_like(dqkv)
    ops.sa_bwd(q, dp, C, TS(dO, 0, C), pk, gpk, fpk, lse, D, TS(dq2, 0, CQ))
    torch.cuda.synchronize()
    assert torch.equal(dq2, dqkv)


@pytest.mark.parametrize("size,arch", [((192, 256), "xresnet34"), ((200, 200), "xresnet34")])
def test_fused_module_against_the_blockwise_path(size, arch):
    """HipDynamicUnet(self_attention=True, act_dtype='bf16'): the fused kernels against the blockwise products of the same library (same
    rounding points: weights and softmax adjoint in bf16, logits / dP fp32): eval logits rel-L2 <= 5e-3, loss within 2e-3 relative, flat gradient
    cosine >= 0.999, attention parameters >= 0.99"""
    import torch.nn.functional as F
    from oracle import unet_oracle as O
    from tests.test_configs_gpu import _normalise_head, _sa_pair
    from unet_amd.model import HipDynamicUnet
    from unet_amd.modules import SelfAttention
    x, y = O.synthetic_batch(2, 3, size[0], size[1], 3)
    ref = _sa_pair(arch, 3, 3, size, 61, x)
    _normalise_head(ref, x[:1])
    w = torch.tensor([0.2, 0.5, 0.3])
    res = {}
    old = SelfAttention.fused
    try:
        for fused in (False, True):
            SelfAttention.fused = fused
            m = HipDynamicUnet(arch, 3, 3, size, self_attention=True, act_dtype="bf16")
            m.load_state_dict(ref.state_dict())
            m.eval()
            with torch.no_grad():
                z = m(x.cuda()).float().cpu()
            m.train()
            loss = m.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().flatten().cpu().double() for n, p in m.named_parameters()}
            res[fused] = (z, loss.item(), grads)
    finally:
        SelfAttention.fused = old
    (z0, l0, g0), (z1, l1, g1) = res[False], res[True]
    rel = ((z1 - z0).norm() / z0.norm()).item()
    cos = F.cosine_similarity(torch.cat(list(g1.values())), torch.cat(list(g0.values())), dim=0).item()
    sa = {n.split("conv2.2.")[1]: F.cosine_similarity(g1[n], g0[n], dim=0).item() for n in g0 if ".conv2.2." in n}
    print(f"fused vs blockwise {arch} {size}: logits rel-L2 {rel:.3e}, loss {l1:.6f} vs {l0:.6f}, gradient cos {cos:.6f}, attention parameters {sa}")
    assert rel <= 5e-3 and abs(l1 - l0) <= 2e-3 * abs(l0)
    assert cos >= 0.999 and len(sa) == 4 and all(v >= 0.99 for v in sa.values()), sa


def test_fused_attention_step_repeats_bit_for_bit():
    """No atomics in the fused backward: two models built from the same state give the same gradient BITS after one training step.
    TrainStep(use_graph=True) on a model with SelfAttention falls back to the eager launch stream with a warning (capturing the spectral-norm
    autograd ends in a segmentation fault of hipStreamEndCapture on this stack: scripts/graph_sa_probe.py) and trains."""
    import warnings
    from oracle import unet_oracle as O
    from unet_amd.model import HipDynamicUnet
    from unet_amd.optimizer import FlatAdam
    from unet_amd.trainer import TrainStep
    torch.manual_seed(5)
    size = (128, 160)                                      # SelfAttention(384) on 16 x 20 = 320 positions
    sd = O.DynamicUnet("xresnet34", 3, 4, size, self_attention=True).state_dict()
    for k in sd:
        if k.endswith("gamma"):
            sd[k] = torch.tensor([0.7])                    # (fastai initialises gamma to 0: the attention branch would carry no signal)
    x, y = O.synthetic_batch(2, 3, size[0], size[1], 4)
    w = torch.tensor([0.1, 0.4, 0.3, 0.2])
    grads = []
    for _ in range(2):
        m = HipDynamicUnet("xresnet34", 3, 4, size, self_attention=True, act_dtype="bf16")
        m.load_state_dict(sd)
        m.train()
        m.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
        torch.cuda.synchronize()
        grads.append(torch.cat([p.grad.detach().flatten() for p in m.parameters()]).cpu())
    assert torch.equal(grads[0], grads[1])
    assert grads[0].abs().max().item() > 0
    m = HipDynamicUnet("xresnet34", 3, 4, size, self_attention=True, act_dtype="bf16")
    m.load_state_dict(sd)
    m.train()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        step = TrainStep(m, FlatAdam(m, [1e-5, 3e-5, 1e-4]), None, 1, use_graph=True)
    assert not step.use_graph and any("SelfAttention" in str(r.message) for r in rec)
    losses = [float(step(x.cuda(), y.cuda()).item()) for _ in range(4)]
    print("SA model, TrainStep(use_graph=True) -> eager:", losses)
    assert step._graph is None and all(l == l for l in losses) and losses[-1] < losses[0]
