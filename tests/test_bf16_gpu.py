"""bf16-storage / fp32-accumulate variant (BASELINE.json configs[1] is quoted as "bf16"; the reference itself is fp32, so fp32 stays
the parity path and THIS file states the bf16 tolerances).

Kernel level: inputs are made bf16-representable, so every product is exact in fp32 and the only differences to an fp32/fp64
reference are the fp32 accumulation order and ONE rounding of the output to bf16 (relative 2^-9 per element, fp32 outputs: none).
Network level, against the fp32 CPU oracle (xresnet34 4->5, the headline geometry):
    eval logits      relative L2 <= 2e-2, max abs <= 5e-2 x logit scale
    argmax masks     >= 99 % of the pixels identical; every differing pixel has an oracle top-2 margin below the measured logit error
    training loss    within 5e-3 relative
    gradients        cosine similarity >= 0.99 for the whole flat gradient and >= 0.97 for every decoder tensor
"""
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)
from tests.util import from_ts  # noqa: E402


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _ts(x_nchw, cs=None, co=0, fill=7.25):
    """NCHW fp32 (bf16-representable) -> bf16 NHWC channel slice on the device"""
    from unet_amd import ops
    from unet_amd.ops import TS
    N, C, H, W = x_nchw.shape
    cs = ops.rupv(C, torch.bfloat16) if cs is None else cs
    buf = torch.full((N, H, W, cs), fill, dtype=torch.float32)
    buf[..., co:co + C] = x_nchw.permute(0, 2, 3, 1)
    if cs > co + C:
        buf[..., co + C:co + ops.rupv(C, torch.bfloat16)] = 0.0
    return TS(buf.to(torch.bfloat16).cuda().contiguous(), co, C)


def _empty(N, H, W, C, cs=None, co=0, dtype=torch.bfloat16, fill=7.25):
    from unet_amd import ops
    from unet_amd.ops import TS
    cs = ops.rupv(C, dtype) if cs is None else cs
    return TS(torch.full((N, H, W, cs), fill, dtype=dtype, device="cuda"), co, C)


def _back(t):
    return t.view().float().permute(0, 3, 1, 2).contiguous().cpu()


CASES = [
    # N, H,  W,  Cin, Cout, ks, stride
    (2, 16, 16, 32, 32, 3, 1),
    (2, 40, 48, 100, 100, 3, 1),      # final ResBlock width (reduction tail 4 of 32), ragged tiles
    (1, 16, 16, 192, 96, 3, 1),
    (1, 8, 8, 512, 1024, 3, 1),       # middle_conv
    (2, 32, 32, 4, 32, 3, 2),         # stem conv0: 4 input channels in an 8-wide buffer
    (2, 26, 26, 64, 128, 3, 2),       # strided ResBlock conv
    (2, 16, 16, 512, 1024, 1, 1),     # PixelShuffle_ICNR 1x1
    (2, 32, 32, 100, 5, 1, 1),        # head
    (1, 24, 40, 96, 192, 3, 1),       # 128 + 64 produced channels: two channel-range launches
    (1, 32, 32, 64, 392, 1, 1),
    # reduction tails of 1..8 channels run tap-folded (4 taps per MFMA; conv_common.h bf16_fold_tail)
    (1, 24, 40, 36, 100, 3, 1),       # fold, two chunks; its dgrad reduces over 100: fold again
    (2, 20, 24, 104, 40, 3, 1),       # tail of exactly 8
    (1, 20, 24, 105, 33, 3, 1),       # tail of 9: not folded; dgrad tail of 1: folded
    (2, 26, 26, 64, 100, 3, 2),       # stride-2 dgrad over 100 channels: four tap sets, reads the UNfolded tail of the same image
    (1, 18, 18, 7, 24, 3, 1),         # everything is tail
]


@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad_bf16(case):
    from unet_amd import ops
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(hash(case) % 997)
    pad = (ks - 1) // 2
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    wb = _bf(w)                                      # what the packed image holds
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), wb.double(), b.double(), stride=stride, padding=pad)
    OH, OW = ref.shape[-2:]
    xt = _ts(x)
    # forward, bf16 output
    yt = _empty(N, OH, OW, Cout)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    ops.conv2d(xt, wp, yt, ks, stride, bias=b.cuda())
    got = _back(yt)
    assert (got.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6, "fwd bf16"
    assert (got.double() - ref).norm().item() <= 3e-3 * ref.norm().item()
    # forward, fp32 output (the logits path): only the accumulation order differs
    if Cout <= 16:
        yf = _empty(N, OH, OW, Cout, dtype=torch.float32)
        ops.conv2d(xt, wp, yf, ks, stride, bias=b.cuda())
        gotf = from_ts(yf)
        assert (gotf.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), "fwd fp32 out"
    # input gradient
    dy = _bf(torch.randn(N, Cout, OH, OW, generator=g))
    dx_ref = torch.nn.grad.conv2d_input(x.shape, wb.double(), dy.double(), stride=stride, padding=pad)
    dyt, dxt = _ts(dy), _empty(N, H, W, Cin)
    ops.conv2d_dgrad(dyt, ops.pack_weights(w.cuda(), 1, dtype=torch.bfloat16), dxt, ks, stride)
    gdx = _back(dxt)
    assert (gdx.double() - dx_ref).abs().max().item() <= 2.0 ** -8 * dx_ref.abs().max().item() + 1e-6, "dgrad"
    # weight (+ bias) gradient: fp32 result
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=stride, padding=pad)
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    db = torch.empty(Cout, device="cuda")
    n = ops.wgrad_workspace(xt, dyt, ks, stride, with_bias=True)
    ops.conv2d_wgrad(xt, dyt, dw, ks, stride, torch.empty(n, device="cuda"), dbias=db)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - dw_ref).abs().max().item() <= 5e-5 * dw_ref.abs().max().item(), "wgrad"
    assert (db.cpu().double() - dy.double().sum((0, 2, 3))).abs().max().item() <= 5e-5 * dy.double().sum((0, 2, 3)).abs().max().item() + 1e-5


@pytest.mark.parametrize("case", [(96, 384, 64, 64, 8), (64, 256, 61, 67, 5), (256, 96, 128, 128, 2), (32, 200, 128, 128, 2)])
@pytest.mark.parametrize("mode", [1, 2])
def test_conv1x1_gemm_kernel_bf16(case, mode):
    from unet_amd import ops
    with ops.tuning(conv1x1_gemm=mode):
        _conv1x1_gemm_case_bf16(case)


def _conv1x1_gemm_case_bf16(case):
    """conv1x1_gemm_kernel<bf16> (variant 8): 1x1 / stride 1 over whole 32-channel chunks as a flat-pixel GEMM; bf16 output within one rounding
    of the fp64 result on bf16-representable operands, fp32 output (y_f32) to 5e-5; gradient form with residual + mask"""
    from unet_amd import ops
    Cin, Cout, H, W, N = case
    g = torch.Generator().manual_seed(Cin + 7 * Cout)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double()) + r.double())
    xt, rt = _ts(x, cs=Cin + 16, co=8), _ts(r)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    yt = _empty(N, H, W, Cout, cs=ops.rupv(Cout, torch.bfloat16) + 16, co=8)
    yf = _empty(N, H, W, Cout, dtype=torch.float32)
    assert ops.conv2d_variant(xt, wp, yt, 1) == 8
    ops.conv2d(xt, wp, yt, 1, bias=b.cuda(), res=rt, relu=True)
    ops.conv2d(xt, wp, yf, 1, bias=b.cuda())
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (_back(yt).double() - ref).abs().max().item() <= 2.0 ** -8 * scale
    pre = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    assert (_back(yf).double() - pre).abs().max().item() <= 5e-5 * pre.abs().max().item()
    assert bool((yt.buf[..., :8] == 7.25).all()) and bool((yt.buf[..., 8 + ops.rupv(Cout, torch.bfloat16):] == 7.25).all())
    if Cout % 32 == 0 and Cin >= 128:
        dy = _bf(torch.randn(N, Cout, H, W, generator=g))
        act = _bf(torch.relu(torch.randn(N, Cin, H, W, generator=g)))
        extra = _bf(torch.randn(N, Cin, H, W, generator=g))
        dref = (torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy.double()) + extra.double()) * (act > 0)
        dxt = _empty(N, H, W, Cin)
        wpd = ops.pack_weights(w.cuda(), 1, dtype=torch.bfloat16)
        dyt = _ts(dy)
        assert ops.conv2d_variant(dyt, wpd, dxt, 1, 1, kind=1) == 8
        ops.conv2d_dgrad(dyt, wpd, dxt, 1, 1, res=_ts(extra), mask=_ts(act))
        torch.cuda.synchronize()
        assert (_back(dxt).double() - dref).abs().max().item() <= 2.0 ** -8 * dref.abs().max().item() + 1e-6


@pytest.mark.parametrize("case", [(2, 8, 8, 512, 512, 3), (1, 16, 16, 1024, 200, 3), (1, 16, 16, 2048, 96, 1), (2, 8, 8, 264, 100, 3)])
def test_conv_splitk_bf16(case):
    """split-K with bf16 storage: fp32 partial slabs, one rounding to bf16 in the reduce kernel -- the same numbers as the unsplit launch up to
    the order of the fp32 partial sums (bf16 output: at most one ulp apart), fp32 logits output equal to 1e-5"""
    from unet_amd import ops
    from unet_amd._lib import lib
    N, H, W, Cin, Cout, ks = case
    g = torch.Generator().manual_seed(sum(case))
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5)
    b = torch.randn(Cout, generator=g)
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=ks // 2) + r.double())
    xt, rt = _ts(x), _ts(r)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    res = {}
    for on in (1, 0):
        _knobs.set_knob("conv_splitk", on)
        try:
            yt = _empty(N, H, W, Cout)
            yf = _empty(N, H, W, Cout, dtype=torch.float32)
            var = ops.conv2d_variant(xt, wp, yt, ks)
            ops.conv2d(xt, wp, yt, ks, bias=b.cuda(), res=rt, relu=True)
            ops.conv2d(xt, wp, yf, ks, bias=b.cuda())
            torch.cuda.synchronize()
        finally:
            _knobs.set_knob("conv_splitk", 1)
        assert (var >= 2000000) == bool(on) or Cout > 128, (case, on, var)
        res[on] = (_back(yt), _back(yf))
    scale = ref.abs().max().item()
    assert (res[1][0].double() - ref).abs().max().item() <= 2.0 ** -8 * scale
    assert (res[1][0] - res[0][0]).abs().max().item() <= 2.0 ** -7 * scale                  # one bf16 ulp where the fp32 sums straddle a tie
    pre = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    assert (res[1][1].double() - pre).abs().max().item() <= 2e-5 * pre.abs().max().item()
    assert (res[1][1] - res[0][1]).abs().max().item() <= 1e-5 * pre.abs().max().item()


@pytest.mark.parametrize("Cout", [100, 112, 97])
def test_conv_bf16_big_tile_shared_odd_tile(Cout):
    """2 x 256 x 256 pixels = 512 blocks of 256 x 128: the 8-pixel-tile wave (variant 321287, the dominant kernel of the bf16 step) with
    7 live output tiles -- the odd one shared between the two waves of a pixel row, 4 pixel tiles each."""
    from unet_amd import ops
    N, H, W, Cin = 2, 256, 256, 100
    g = torch.Generator().manual_seed(Cout)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    ref = F.relu(F.conv2d(x, _bf(w), b, padding=1) + r)
    xt, rt = _ts(x), _ts(r, cs=144, co=16)
    yt = _empty(N, H, W, Cout, cs=160, co=32)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    assert ops.conv2d_variant(xt, wp, yt, 3, 1) == 321287
    ops.conv2d(xt, wp, yt, 3, 1, bias=b.cuda(), res=rt, relu=True)
    got = _back(yt)
    assert (got - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-5
    assert (got - ref).norm().item() <= 3e-3 * ref.norm().item()
    full = yt.buf.float().cpu()
    assert bool((full[..., :32] == 7.25).all()) and bool((full[..., 32 + ops.rupv(Cout, torch.bfloat16):] == 7.25).all()), "wrote outside the slice"


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (Cin, Cout, H, W): 16-wide channel tiles per 128-block / reduction tail
    (32, 72, 256, 256),       # 5 tiles (shared third tile), one chunk
    (96, 96, 250, 270),       # 6 tiles, ragged image edges (out-of-image halo items and pixels)
    (100, 100, 256, 256),     # 7 tiles, tap-folded reduction tail
    (112, 128, 256, 256),     # 8 tiles, zero-padded tail chunk (16 live channels of 32)
    (40, 228, 256, 256),      # two launches: a full block of 8 tiles + a narrow block of 7; folded tail of 8 channels
    (64, 160, 256, 256),      # 8 tiles + a 2-tile block (runs as 5 tiles over the zero filters of the padded image)
    (256, 128, 32, 32, 16),   # a deep stage: 64 blocks of 256 pixels, the reduction split in two on top (fp32 slabs + reduce kernel)
    (64, 64, 64, 64, 16),     # 64-wide channel block: 4 tiles, two per wave
    (32, 40, 128, 64, 4),     # 3 tiles: one per wave + a shared one
    (24, 32, 64, 64, 16),     # 32-wide block: one tile per wave
    (32, 16, 64, 96, 8),      # a single tile shared by the two waves of a pixel row
    (100, 72, 32, 24, 16),    # 16-pixel-wide patches (outputs 16..31 pixels wide), ragged second patch column, folded tail
    (256, 512, 16, 16, 16),   # a 16 x 16 stage: 16 patches x 4 channel blocks
])
def test_conv_bf16_t256_kernel(case):
    """conv_bf16_t256_kernel (the 256-pixel tile of the large 3x3 layers, variant ...7): forward with bias + residual + ReLU into a channel
    slice, fp32 output, and the input gradient (filter slabs walked backwards) with residual + mask -- against fp64 on the same bf16 values.
    The input is a slice of a wider buffer whose neighbouring channels are NOT zero: a tail chunk must not read them."""
    from unet_amd import ops
    Cin, Cout, H, W = case[:4]
    N = case[4] if len(case) > 4 else 2
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wb = _bf(w)
    b = torch.randn(Cout, generator=g)
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    ref = F.relu(F.conv2d(x.double(), wb.double(), b.double(), padding=1) + r.double())
    ci, co = ops.rupv(Cin, torch.bfloat16), ops.rupv(Cout, torch.bfloat16)
    xt = _ts(x, cs=ci + 48, co=16)
    rt = _ts(r, cs=co + 16, co=16)
    yt = _empty(N, H, W, Cout, cs=co + 40, co=32)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    assert ops.conv2d_variant(xt, wp, yt, 3, 1) % 10 in (6, 7)          # the 256-pixel tile (7: a large layer, 6: narrow / small)
    ops.conv2d(xt, wp, yt, 3, 1, bias=b.cuda(), res=rt, relu=True)
    got = _back(yt)
    assert (got.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-5
    full = yt.buf.float().cpu()
    assert bool((full[..., :32] == 7.25).all()) and bool((full[..., 32 + co:] == 7.25).all()), "wrote outside the slice"
    # fp32 output (the logits form), no epilogue
    yf = _empty(N, H, W, Cout, dtype=torch.float32)
    ops.conv2d(xt, wp, yf, 3, 1)
    ref0 = F.conv2d(x.double(), wb.double(), padding=1)
    assert (from_ts(yf).cpu().double() - ref0).abs().max().item() <= 5e-5 * ref0.abs().max().item() + 1e-6
    # input gradient
    dy = _bf(torch.randn(N, Cout, H, W, generator=g))
    m = _bf(torch.randn(N, Cin, H, W, generator=g))
    rr = _bf(torch.randn(N, Cin, H, W, generator=g))
    dref = (torch.nn.grad.conv2d_input(x.shape, wb.double(), dy.double(), padding=1) + rr.double()) * (m > 0)
    dxt = _empty(N, H, W, Cin, cs=ci + 8, co=8)
    wpd = ops.pack_weights(w.cuda(), 1, dtype=torch.bfloat16)
    dyt = _ts(dy, cs=co + 24, co=8)
    if not (H == 16 and Cin < 512):          # (the 16 x 16 gradient launch over 256 channels has 32 blocks: generic kernel)
        assert ops.conv2d_variant(dyt, wpd, dxt, 3, 1, kind=1) % 10 in (6, 7)
    ops.conv2d_dgrad(dyt, wpd, dxt, 3, 1, res=_ts(rr), mask=_ts(m, cs=ci + 16, co=16))
    assert (_back(dxt).double() - dref).abs().max().item() <= 2.0 ** -8 * dref.abs().max().item() + 1e-5
    # weight + bias gradient (wgrad_bf16_k4_kernel: 3x3 / stride 1 / 32-wide pixel tiles; blocks of 3 or 4 output-channel tiles, image
    # edges through the buffer range check and the two column flags, operands in slices with non-zero neighbours), fp32 result
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    db = torch.empty(Cout, device="cuda")
    n = ops.wgrad_workspace(xt, dyt, 3, 1, with_bias=True)
    ops.conv2d_wgrad(xt, dyt, dw, 3, 1, torch.empty(n, device="cuda"), dbias=db)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - dw_ref).abs().max().item() <= 5e-5 * dw_ref.abs().max().item(), "wgrad"
    dbr = dy.double().sum((0, 2, 3))
    assert (db.cpu().double() - dbr).abs().max().item() <= 5e-5 * dbr.abs().max().item() + 1e-5


@pytest.mark.parametrize("K,Cout", [(5, 100), (8, 24)])
def test_conv1x1_small_reduction_kernel_bf16(K, Cout):
    """conv1x1_smallk_kernel with bf16 storage (variant id 9): gradient form with residual + mask, forward form with bias + ReLU"""
    from unet_amd import ops
    N, H, W = 2, 37, 29
    g = torch.Generator().manual_seed(K * 100 + Cout)
    wf = torch.randn(K, Cout, 1, 1, generator=g)
    wb = _bf(wf)
    dy = _bf(torch.randn(N, K, H, W, generator=g))
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    act = _bf(F.relu(torch.randn(N, Cout, H, W, generator=g)))
    ref = (torch.nn.grad.conv2d_input((N, Cout, H, W), wb.double(), dy.double()) + r.double()) * (act > 0)
    co = ops.rupv(Cout, torch.bfloat16)
    dxt = _empty(N, H, W, Cout, cs=co + 24, co=16)
    dyt = _ts(dy, cs=24, co=8)
    wpd = ops.pack_weights(wf.cuda(), 1, dtype=torch.bfloat16)
    assert ops.conv2d_variant(dyt, wpd, dxt, 1, 1, kind=1) == 9
    ops.conv2d_dgrad(dyt, wpd, dxt, 1, 1, res=_ts(r), mask=_ts(act, cs=co + 8, co=8))
    assert (_back(dxt).double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6
    full = dxt.buf.float().cpu()
    assert bool((full[..., :16] == 7.25).all()) and bool((full[..., 16 + co:] == 7.25).all()), "wrote outside the slice"
    w = torch.randn(Cout, K, 1, 1, generator=g)
    b = torch.randn(Cout, generator=g)
    x = _bf(torch.randn(N, K, H, W, generator=g))
    yt = _empty(N, H, W, Cout)
    ops.conv2d(_ts(x), ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16), yt, 1, 1, bias=b.cuda(), relu=True)
    ref2 = F.relu(F.conv2d(x.double(), _bf(w).double(), b.double()))
    assert (_back(yt).double() - ref2).abs().max().item() <= 2.0 ** -8 * ref2.abs().max().item() + 1e-6


@pytest.mark.parametrize("case", [(2, 37, 29, 100, 5), (1, 64, 64, 96, 2), (2, 16, 16, 128, 16), (1, 20, 20, 36, 3), (3, 5, 5, 9, 7)])
@pytest.mark.parametrize("f32_out", [True, False])
def test_conv1x1_head_kernel_bf16(case, f32_out):
    """conv1x1_head_kernel with bf16 storage (variant id 11): fp32 logits (the head's own form: unet_conv_desc.y_f32) and bf16 output; exact
    products, fp32 accumulation; the same bits as the implicit-GEMM kernel (unet_tuning.conv_head1x1 = 0)"""
    from unet_amd import ops
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), _bf(w).double(), b.double())
    xt = _ts(x, cs=ops.rupv(Cin, torch.bfloat16) + 16, co=8)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    odt = torch.float32 if f32_out else torch.bfloat16
    outs = []
    for on in (1, 2, 0):                      # 2: the other pixel-tile count per trip
        with _knobs.tuning(conv_head1x1=on, conv_splitk=0):       # (a split reduction is another chain: the planner cuts 128 channels on a small grid)
            yt = _empty(N, H, W, Cout, cs=ops.rupv(Cout, odt) + 16, co=8, dtype=odt)
            assert (ops.conv2d_variant(xt, wp, yt, 1, 1) == 11) == bool(on)
            ops.conv2d(xt, wp, yt, 1, 1, bias=b.cuda())
            torch.cuda.synchronize()
        full = yt.buf.float().cpu()
        assert bool((full[..., :8] == 7.25).all()) and bool((full[..., 8 + ops.rupv(Cout, odt):] == 7.25).all()), "wrote outside the slice"
        outs.append(_back(yt).double())
    scale = ref.abs().max().item()
    assert (outs[0] - ref).abs().max().item() <= (1e-5 if f32_out else 2.0 ** -8) * scale + 1e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "not the bits of the implicit-GEMM kernel"


@pytest.mark.parametrize("case", [(2, 64, 64, 4, 32, 2), (1, 37, 29, 3, 32, 2), (2, 18, 22, 7, 24, 1), (1, 33, 31, 8, 32, 1), (5, 512, 512, 4, 32, 2)])
def test_conv3x3_small_cin_kernel_bf16(case):
    """conv3x3_smallcin_kernel with bf16 storage (variant id 10; the stem's first conv): exact products, fp32 accumulation, one rounding of
    the output; and the implicit-GEMM kernel on the same operands (unet_tuning.conv_smallcin = 0) within two output roundings"""
    from unet_amd import ops
    N, H, W, Cin, Cout, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), _bf(w).double(), b.double(), stride=stride, padding=1))
    OH, OW = ref.shape[2:]
    xt = _ts(x, cs=24, co=8)
    wp = ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16)
    outs = []
    big = N * OH * OW >= 1 << 18
    if not big:                       # bf16 takes the kernel from 2^18 output pixels on (a single tile is faster on MFMA) ...
        yt = _empty(N, OH, OW, Cout, cs=Cout + 24, co=16)
        assert ops.conv2d_variant(xt, wp, yt, 3, stride) != 10
    else:                             # ... counted with the PLANNED batch: a batch-invariant plan (plan_batch = 1) never takes it
        with _knobs.tuning(plan_batch=1):
            assert ops.conv2d_variant(xt, wp, _empty(N, OH, OW, Cout), 3, stride) != 10
    for on in (1, 0):
        with _knobs.tuning(conv_smallcin=on, plan_batch=0 if big else 4096):
            yt = _empty(N, OH, OW, Cout, cs=Cout + 24, co=16)
            assert (ops.conv2d_variant(xt, wp, yt, 3, stride) == 10) == bool(on)
            ops.conv2d(xt, wp, yt, 3, stride, bias=b.cuda(), relu=True)
            torch.cuda.synchronize()
        full = yt.buf.float().cpu()
        assert bool((full[..., :16] == 7.25).all()) and bool((full[..., 16 + Cout:] == 7.25).all()), "wrote outside the slice"
        outs.append(_back(yt).double())
    scale = ref.abs().max().item()
    assert (outs[0] - ref).abs().max().item() <= 2.0 ** -8 * scale + 1e-6
    assert (outs[0] - outs[1]).abs().max().item() <= 2.0 ** -7 * scale + 1e-6


def test_conv_epilogue_slices_residual_relu_mask_bf16():
    """channel-sliced operands (concat elimination), bias + residual + ReLU forward epilogue, residual + ReLU-mask dgrad epilogue"""
    from unet_amd import ops
    N, H, W, Cin, Cout = 2, 24, 40, 40, 96
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wb = _bf(w)
    b = torch.randn(Cout, generator=g)
    r = _bf(torch.randn(N, Cout, H, W, generator=g))
    ref = F.relu(F.conv2d(x.double(), wb.double(), b.double(), padding=1) + r.double())
    xt, rt = _ts(x, cs=64, co=8), _ts(r, cs=128, co=16)
    yt = _empty(N, H, W, Cout, cs=160, co=32)
    ops.conv2d(xt, ops.pack_weights(w.cuda(), 0, dtype=torch.bfloat16), yt, 3, 1, bias=b.cuda(), res=rt, relu=True)
    got = _back(yt)
    assert (got.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6
    full = yt.buf.float().cpu()
    assert bool((full[..., :32] == 7.25).all()) and bool((full[..., 128:] == 7.25).all()), "wrote outside the output slice"
    dy = _bf(torch.randn(N, Cout, H, W, generator=g))
    m = _bf(torch.randn(N, Cin, H, W, generator=g))
    rr = _bf(torch.randn(N, Cin, H, W, generator=g))
    dref = (torch.nn.grad.conv2d_input(x.shape, wb.double(), dy.double(), padding=1) + rr.double()) * (m > 0)
    dxt = _empty(N, H, W, Cin, cs=48, co=8)
    ops.conv2d_dgrad(_ts(dy), ops.pack_weights(w.cuda(), 1, dtype=torch.bfloat16), dxt, 3, 1, res=_ts(rr), mask=_ts(m, cs=56, co=16))
    assert (_back(dxt).double() - dref).abs().max().item() <= 2.0 ** -8 * dref.abs().max().item() + 1e-6


def test_elementwise_twins_bf16():
    """BatchNorm statistics / apply / backward, pooling, shuffle + blur and their adjoints, slice copy with bf16 tensors: equal to
    the fp32 entry point of the same name on the same (bf16-representable) values, up to one output rounding"""
    from unet_amd import ops
    from unet_amd.ops import TS
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 12, 20, 64
    x = _bf(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)

    def f32(t):
        return TS(t.permute(0, 2, 3, 1).contiguous().cuda(), 0, t.shape[1])
    tol = lambda ref: 2.0 ** -8 * ref.abs().max().item() + 1e-6
    # statistics: identical partial sums (the inputs are the same numbers, the arithmetic is fp32 in both)
    rows = ops.bn_stats_rows(N * H * W)
    pa, pb = torch.zeros(2 * rows * C, device="cuda"), torch.zeros(2 * rows * C, device="cuda")
    ops.bn_stats(_ts(x), pa); ops.bn_stats(f32(x), pb)
    assert torch.equal(pa, pb)
    sc, sh = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda()
    x2 = _bf(torch.randn(N, C, H, W, generator=g))
    ya, yb = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ops.affine_act(_ts(x), ya, sc, sh, x2=_ts(x2), relu=True); ops.affine_act(f32(x), yb, sc, sh, x2=f32(x2), relu=True)
    assert (_back(ya) - from_ts(yb)).abs().max().item() <= tol(from_ts(yb))
    # BatchNorm backward
    mean, invstd = x.mean((0, 2, 3)).cuda(), (1.0 / (x.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()).cuda()
    dout = _bf(torch.randn(N, C, H, W, generator=g)); out = _bf(torch.randn(N, C, H, W, generator=g))
    ops.bn_bwd_reduce(_ts(dout), _ts(out), _ts(x), mean, invstd, pa); ops.bn_bwd_reduce(f32(dout), f32(out), f32(x), mean, invstd, pb)
    # (the bf16 entry point reduces 8 channels per thread: the same fp32 terms, grouped into the partial rows differently)
    sa, sb = pa.view(2, rows, C).double().sum(1), pb.view(2, rows, C).double().sum(1)
    assert (sa - sb).abs().max().item() <= 1e-5 * sb.abs().max().item()
    c1, c2, gam = torch.randn(C, generator=g).cuda() * 0.1, torch.randn(C, generator=g).cuda() * 0.1, torch.rand(C, generator=g).cuda() + 0.5
    da, db = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ga, gb = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ops.bn_bwd_apply(_ts(dout), _ts(out), _ts(x), mean, invstd, gam, c1, c2, da, gout=ga)
    ops.bn_bwd_apply(f32(dout), f32(out), f32(x), mean, invstd, gam, c1, c2, db, gout=gb)
    assert (_back(da) - from_ts(db)).abs().max().item() <= tol(from_ts(db)) and torch.equal(_back(ga), from_ts(gb))
    # max pool (values exact: selection only) + adjoint, average pool
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    ia, ib = torch.zeros(N * OH * OW * C, dtype=torch.uint8, device="cuda"), torch.zeros(N * OH * OW * C, dtype=torch.uint8, device="cuda")
    pa_, pb_ = _empty(N, OH, OW, C), _empty(N, OH, OW, C, dtype=torch.float32)
    ops.maxpool(_ts(x), pa_, ia); ops.maxpool(f32(x), pb_, ib)
    assert torch.equal(_back(pa_), from_ts(pb_)) and torch.equal(ia, ib)
    dp = _bf(torch.randn(N, C, OH, OW, generator=g))
    ma, mb = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ops.maxpool_bwd(_ts(dp), ia, ma); ops.maxpool_bwd(f32(dp), ib, mb)
    assert (_back(ma) - from_ts(mb)).abs().max().item() <= tol(from_ts(mb))
    aa, ab = _empty(N, H // 2, W // 2, C), _empty(N, H // 2, W // 2, C, dtype=torch.float32)
    ops.avgpool(_ts(x), aa); ops.avgpool(f32(x), ab)
    assert (_back(aa) - from_ts(ab)).abs().max().item() <= tol(from_ts(ab))
    dav = _bf(torch.randn(N, C, H // 2, W // 2, generator=g))
    va, vb = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ops.avgpool_bwd(_ts(dav), va); ops.avgpool_bwd(f32(dav), vb)
    assert torch.equal(_back(va), from_ts(vb))                  # x 0.25: exact in bf16
    # shuffle + blur into a channel slice of a concat buffer, and the adjoint with the fused ReLU mask
    Cu = C // 4
    yc = _bf(torch.relu(torch.randn(N, C, H, W, generator=g)))
    Xa, Xb = _empty(N, 2 * H, 2 * W, Cu, cs=40, co=8), _empty(N, 2 * H, 2 * W, Cu, cs=40, co=8, dtype=torch.float32)
    ops.shuffle_blur(_ts(yc), Xa, True); ops.shuffle_blur(f32(yc), Xb, True)
    assert (_back(Xa) - from_ts(Xb)).abs().max().item() <= tol(from_ts(Xb))
    assert bool((Xa.buf.float()[..., :8] == 7.25).all()) and bool((Xa.buf.float()[..., 24:] == 7.25).all())
    dX = _bf(torch.randn(N, Cu, 2 * H, 2 * W, generator=g))
    sa_, sb_ = _empty(N, H, W, C), _empty(N, H, W, C, dtype=torch.float32)
    ops.shuffle_blur_bwd(_ts(dX), _ts(yc), sa_, True); ops.shuffle_blur_bwd(f32(dX), f32(yc), sb_, True)
    assert (_back(sa_) - from_ts(sb_)).abs().max().item() <= tol(from_ts(sb_))
    # accumulate-copy
    acc_a, acc_b = _ts(x2), f32(x2)
    ops.copy_slice(_ts(x), acc_a, accumulate=True); ops.copy_slice(f32(x), acc_b, accumulate=True)
    assert (_back(acc_a) - from_ts(acc_b)).abs().max().item() <= tol(from_ts(acc_b))
    # the fp32-only entry points refuse bf16 tensors instead of misreading them
    with pytest.raises(Exception, match="bf16"):
        ops.colsum(_ts(x), torch.zeros(C, device="cuda"), torch.zeros(ops.colsum_workspace(N * H * W, C), device="cuda"))


@pytest.fixture(scope="module")
def nets():
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    ref = O.DynamicUnet("xresnet34", 4, 5, (256, 256))
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    x, y = O.synthetic_batch(2, 4, 256, 256, 5)
    ref.eval()
    with torch.no_grad():
        s = ref(x).abs().max().item() / 4.0
        ref.layers[-1][0].weight.div_(s); ref.layers[-1][0].bias.div_(s)        # logits O(1)
    m16 = HipDynamicUnet("xresnet34", 4, 5, (256, 256), act_dtype="bf16")
    m16.load_state_dict(ref.state_dict())
    return ref, m16, x, y


def test_bf16_network_eval_against_the_fp32_oracle(nets):
    ref, m16, x, _ = nets
    ref.eval(); m16.eval()
    with torch.no_grad():
        z32 = ref(x)
        probs, amax = m16.predict_probs(x.cuda())
        z = m16(x.cuda()).cpu()
    assert z.dtype == torch.float32
    scale = z32.abs().max().item()
    err, rel = (z - z32).abs().max().item(), ((z - z32).norm() / z32.norm()).item()
    agree = (amax.cpu() == z32.argmax(1)).float().mean().item()
    print(f"bf16 eval: logit scale {scale:.2f} max err {err:.3e} rel-L2 {rel:.3e} mask agreement {agree:.5f}")
    assert rel <= 2e-2 and err <= 5e-2 * scale
    assert agree >= 0.99
    diff = amax.cpu() != z32.argmax(1)
    top2 = z32.topk(2, dim=1).values
    assert bool(((top2[:, 0] - top2[:, 1])[diff] <= 2 * err).all()), "a mask pixel differs where the oracle's margin exceeds the bf16 logit error"
    # probabilities: a logit error e moves a softmax weight by at most e / 4 ... e / 2 (two- vs many-way ties); measured 2.9e-2 .. 3.2e-2
    # depending on the summation order of the fp32 partial sums (split-K on the deep stages) in front of the bf16 roundings
    assert (probs.cpu() - torch.softmax(z32, 1)).abs().max().item() <= min(4e-2, err)


def test_bf16_training_step_against_the_fp32_oracle(nets):
    ref, m16, x, y = nets
    w = torch.tensor([0.1, 0.3, 0.2, 0.25, 0.15])
    ref.train(); m16.train()
    for p in ref.parameters():
        p.grad = None
    l32 = O.CrossEntropyLossFlat(weight=w)(ref(x), y)
    l32.backward()
    loss = m16.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert m16.flat_grad.dtype == torch.float32 and m16.flat_param.dtype == torch.float32
    assert abs(loss.item() - l32.item()) <= 5e-3 * abs(l32.item()), (loss.item(), l32.item())
    gh = torch.cat([p.grad.flatten() for p in m16.parameters()]).cpu().double()
    gr = torch.cat([q.grad.flatten() for q in ref.parameters()]).double()
    cos = F.cosine_similarity(gh, gr, dim=0).item()
    rel = ((gh - gr).norm() / gr.norm()).item()
    worst = ("", 1.0)
    for (n, p), (_, q) in zip(m16.named_parameters(), ref.named_parameters()):
        if int(n.split(".")[1]) >= 4 and q.grad.numel() >= 64 and q.grad.abs().max() > 0:
            c = F.cosine_similarity(p.grad.flatten().cpu().double(), q.grad.flatten().double(), dim=0).item()
            if c < worst[1]:
                worst = (n, c)
    print(f"bf16 train step: loss {loss.item():.5f} vs {l32.item():.5f}; whole gradient cos {cos:.5f} rel-L2 {rel:.3e}; worst decoder tensor {worst}")
    assert cos >= 0.99 and worst[1] >= 0.97


def test_bf16_tiles_not_divisible_by_32():
    """the reference's shipped tile size is 400 (params_and_main.py:36): ceil-mode average pools and nearest resizes in the decoder.  bf16 storage
    ran only on /32 tiles before round 3 (no bf16 resize kernels); 3-band 208 x 176 tiles, xresnet34, eval + one training step against the fp32 oracle"""
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(2)
    ref = O.DynamicUnet("xresnet34", 3, 3, (208, 176))
    O.randomize_bn_and_zero_gammas(ref, seed=3)
    x, y = O.synthetic_batch(2, 3, 208, 176, 3)
    ref.eval()
    with torch.no_grad():
        s = ref(x).abs().max().item() / 4.0
        ref.layers[-1][0].weight.div_(s); ref.layers[-1][0].bias.div_(s)
    m16 = HipDynamicUnet("xresnet34", 3, 3, (208, 176), act_dtype="bf16")
    m16.load_state_dict(ref.state_dict())
    m16.eval()
    with torch.no_grad():
        z32 = ref(x)
        _, amax = m16.predict_probs(x.cuda())
        z = m16(x.cuda()).cpu()
    rel = ((z - z32).norm() / z32.norm()).item()
    agree = (amax.cpu() == z32.argmax(1)).float().mean().item()
    print(f"bf16 208x176 eval: rel-L2 {rel:.3e} mask agreement {agree:.5f}")
    assert rel <= 2e-2 and agree >= 0.99
    w = torch.tensor([0.2, 0.5, 0.3])
    ref.train(); m16.train()
    l32 = O.CrossEntropyLossFlat(weight=w)(ref(x), y)
    l32.backward()
    loss = m16.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l32.item()) <= 5e-3 * abs(l32.item())
    gh = torch.cat([p.grad.flatten() for p in m16.parameters()]).cpu().double()
    gr = torch.cat([q.grad.flatten() for q in ref.parameters()]).double()
    cos = F.cosine_similarity(gh, gr, dim=0).item()
    print(f"bf16 208x176 train step: loss {loss.item():.5f} vs {l32.item():.5f}, gradient cos {cos:.5f}")
    assert cos >= 0.99


@pytest.mark.parametrize("size", [(200, 200), (256, 192)])
def test_bf16_self_attention_against_the_fp32_oracle(size):
    """the reference's shipped configuration has self_attention = True (params_and_main.py:83): SelfAttention(384) in bf16 storage -- fused QKV,
    attention logits and the gradient of the attention weights in fp32, everything else bf16.  200 x 200 tiles put 25 x 25 = 625 positions on the
    attention stage (not a multiple of 8: padded rows); eval and one training step against the fp32 oracle with the conditioned fixture of
    tests/test_configs_gpu.py"""
    from tests.test_configs_gpu import _normalise_head, _sa_pair
    from unet_amd.model import HipDynamicUnet
    x, y = O.synthetic_batch(2, 3, size[0], size[1], 3)
    ref = _sa_pair("xresnet34", 3, 3, size, 61, x)
    _normalise_head(ref, x[:1])
    m16 = HipDynamicUnet("xresnet34", 3, 3, size, self_attention=True, act_dtype="bf16")
    r = m16.load_state_dict(ref.state_dict())
    assert not r.missing_keys and not r.unexpected_keys
    ref.eval(); m16.eval()
    with torch.no_grad():
        z32 = ref(x)
        _, amax = m16.predict_probs(x.cuda())
        z = m16(x.cuda()).cpu()
    rel = ((z - z32).norm() / z32.norm()).item()
    agree = (amax.cpu() == z32.argmax(1)).float().mean().item()
    print(f"bf16 + SA {size} eval: rel-L2 {rel:.3e} mask agreement {agree:.5f}")
    assert rel <= 3e-2 and agree >= 0.98
    w = torch.tensor([0.2, 0.5, 0.3])
    ref.train(); m16.train()
    for p in ref.parameters():
        p.grad = None
    l32 = O.CrossEntropyLossFlat(weight=w)(ref(x), y)
    l32.backward()
    loss = m16.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l32.item()) <= 1e-2 * abs(l32.item()), (loss.item(), l32.item())
    gh = torch.cat([p.grad.flatten() for p in m16.parameters()]).cpu().double()
    gr = torch.cat([q.grad.flatten() for q in ref.parameters()]).double()
    cos = F.cosine_similarity(gh, gr, dim=0).item()
    sa = {n.split("conv2.2.")[1]: F.cosine_similarity(p.grad.flatten().cpu().double(), q.grad.flatten().double(), dim=0).item()
          for (n, p), (_, q) in zip(m16.named_parameters(), ref.named_parameters()) if ".conv2.2." in n}
    print(f"bf16 + SA {size} train step: loss {loss.item():.5f} vs {l32.item():.5f}, gradient cos {cos:.5f}, attention parameters {sa}")
    assert cos >= 0.98 and len(sa) == 4 and all(v >= 0.9 for v in sa.values()), sa


def test_bf16_training_follows_the_fp32_hip_path():
    """ten fit steps from the same initial weights: the bf16-storage model's loss curve stays within 2 % of the fp32 HIP path and
    goes down; master weights stay fp32 (updates far below one bf16 ulp are not lost)"""
    from unet_amd.model import HipDynamicUnet
    from unet_amd.optimizer import FlatAdam
    from unet_amd.trainer import TrainStep
    torch.manual_seed(4)
    sd = O.DynamicUnet("xresnet18", 4, 5, (128, 128)).state_dict()
    x, y = O.synthetic_batch(4, 4, 128, 128, 5)
    y = (x[:, 0] * 4.99).long().clamp(0, 4)            # a learnable target
    curves = []
    for dt in ("f32", "bf16"):
        m = HipDynamicUnet("xresnet18", 4, 5, (128, 128), act_dtype=dt)
        m.load_state_dict(sd)
        m.train()
        step = TrainStep(m, FlatAdam(m, [1e-5, 3e-5, 1e-4]), None, 1)
        curves.append([float(step(x.cuda(), y.cuda()).item()) for _ in range(10)])
        if dt == "bf16":
            w0 = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("weight") and v.dim() == 4][:3])
            w1 = torch.cat([p.detach().flatten().cpu() for n, p in m.named_parameters() if n.endswith("weight") and p.dim() == 4][:3])
            assert 0 < (w1 - w0).abs().max().item() < 0.01
    a, b = curves
    print("fp32 losses", [f"{v:.4f}" for v in a], "bf16 losses", [f"{v:.4f}" for v in b])
    assert b[-1] < b[0] and all(abs(u - v) <= 2e-2 * abs(u) for u, v in zip(a, b))


def test_bf16_regression_step_follows_the_fp32_hip_path():
    """enable_regression (n_out = 1, MSE on float targets; reference train.py:137-138,189-193) with bf16 storage: loss equal to the fp32 HIP
    path within bf16 rounding, flat gradient cosine >= 0.99"""
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(3)
    ref = O.DynamicUnet("xresnet18", 4, 1, (64, 64))
    O.randomize_bn_and_zero_gammas(ref, seed=4)
    a = HipDynamicUnet("xresnet18", 4, 1, (64, 64)); a.load_state_dict(ref.state_dict()); a.train()
    b = HipDynamicUnet("xresnet18", 4, 1, (64, 64), act_dtype="bf16"); b.load_state_dict(ref.state_dict()); b.train()
    x, _ = O.synthetic_batch(2, 4, 64, 64, 2)
    y = torch.rand(2, 64, 64, generator=torch.Generator().manual_seed(10)) * 3
    la = a.forward_loss_backward(x.cuda(), y.cuda(), reg_kind="mse").item()
    lb = b.forward_loss_backward(x.cuda(), y.cuda(), reg_kind="mse").item()
    assert abs(la - lb) <= 2e-2 * max(1.0, abs(la)), (la, lb)
    ga, gb = a.flat_grad.double(), b.flat_grad.double()
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos >= 0.99, cos


def test_bf16_xresnet34_deep_ragged_decoder_widths():
    """xresnet34_deep with bf16 storage: the last UnetBlock up-samples 140 channels and the dense merge follows 102 -- not multiples of the
    8-channel vector, so the concat buffers carry a gap and the consumers' filters zero rows / columns for it (modules._ConvExec.set_gaps).
    Loss equal to the fp32 HIP path within bf16 rounding; the gradients of the four gapped layers (filters AND biases: live rows / columns
    scattered back from the gapped gradient) agree to cosine >= 0.999; flat gradient cosine >= 0.97 (measured 0.980: this encoder's two
    extra stages are 4 x 4 and 2 x 2 pixels at 256 x 256 -- BatchNorm over 8..32 values makes the encoder's small gradients noisy in bf16,
    the standard xresnet34 reaches 0.99996); eval masks >= 98 % identical."""
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(5)
    S = 256
    ref = O.DynamicUnet("xresnet34_deep", 4, 5, (S, S))
    O.randomize_bn_and_zero_gammas(ref, seed=6)
    a = HipDynamicUnet("xresnet34_deep", 4, 5, (S, S)); a.load_state_dict(ref.state_dict()); a.train()
    b = HipDynamicUnet("xresnet34_deep", 4, 5, (S, S), act_dtype="bf16"); b.load_state_dict(ref.state_dict()); b.train()
    assert (b.up_c, b.up_off, b.cat_c, b.cat_p) == (102, 104, 106, 108) and (b.layers[9].cu, b.layers[9].cu_off) == (140, 144)
    x, y = O.synthetic_batch(2, 4, S, S, 5, seed=7)
    w = torch.full((5,), 0.2, device="cuda")
    la = a.forward_loss_backward(x.cuda(), y.cuda(), w).item()
    lb = b.forward_loss_backward(x.cuda(), y.cuda(), w).item()
    assert abs(la - lb) <= 2e-2 * max(1.0, abs(la)), (la, lb)
    ga, gb = a.flat_grad.double(), b.flat_grad.double()
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos >= 0.97, cos
    seen = 0
    for (n, p), (_, q) in zip(b.named_parameters(), a.named_parameters()):
        if n.startswith(("layers.9.conv1.", "layers.13.", "layers.14.")):
            pg, qg = p.grad.double().flatten(), q.grad.double().flatten()
            c = (pg @ qg / (pg.norm() * qg.norm() + 1e-30)).item()
            assert c >= 0.999, (n, c)
            seen += 1
    assert seen == 8
    a.eval(); b.eval()
    with torch.no_grad():
        ma, mb = a.predict_probs(x.cuda())[1], b.predict_probs(x.cuda())[1]
    assert (ma == mb).float().mean().item() >= 0.98
