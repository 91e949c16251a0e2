"""Per-kernel parity: every C-ABI entry point against the plain torch-CPU fp32 op it replaces.

Tolerances: fp32 everywhere.  Convolutions reduce over up to 9*1024 products in a
different order than oneDNN, so they are compared at rtol 2e-4 of the tensor's max
magnitude; data movement / pooling / masks are bit exact.
"""
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from unet_amd import ops as _ops
    return _ops


from tests.util import assert_close, empty_ts, from_ts, outside_untouched, to_ts  # noqa: E402

CONV_CASES = [
    # N, H,  W,  Cin, Cout, ks, stride
    (2, 16, 16, 32, 32, 3, 1),
    (1, 32, 32, 64, 64, 3, 1),
    (2, 40, 48, 100, 100, 3, 1),      # final ResBlock width, ragged tiles
    (1, 16, 16, 192, 96, 3, 1),
    (1, 8, 8, 512, 1024, 3, 1),       # middle_conv, tiny spatial
    (2, 13, 13, 128, 128, 3, 1),      # odd size (400-px tiles path)
    (2, 32, 32, 4, 32, 3, 2),         # stem conv0
    (1, 16, 16, 3, 32, 3, 2),         # 3-channel stem (channel stride 4)
    (2, 26, 26, 64, 128, 3, 2),       # strided ResBlock conv
    (1, 25, 25, 64, 128, 3, 2),       # odd input, stride 2
    (2, 16, 16, 512, 1024, 1, 1),     # PixelShuffle_ICNR 1x1
    (1, 64, 64, 96, 384, 1, 1),
    (2, 32, 32, 100, 5, 1, 1),        # head
    (1, 32, 32, 99, 2, 1, 1),         # 3-channel config head (99 = 96 + 3)
    (1, 16, 16, 64, 128, 1, 1),       # identity-path 1x1
    (1, 24, 40, 96, 192, 3, 1),       # 192 = 128 + 64 produced channels: issued as two channel-range launches
    (2, 16, 16, 32, 136, 3, 1),       # 128 + 8
    (1, 32, 32, 64, 392, 1, 1),       # 3 * 128 + 8 (xresnet50-width decoder)
]


def _conv_ref(x, w, b, ks, stride):
    return F.conv2d(x, w, b, stride=stride, padding=(ks - 1) // 2)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(ops, case):
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = _conv_ref(x, w, b, ks, stride)
    OH, OW = ref.shape[-2:]
    xt = to_ts(x)
    yt = empty_ts(N, OH, OW, Cout)
    wp = ops.pack_weights(w.cuda(), 0)
    ops.conv2d(xt, wp, yt, ks, stride, bias=b.cuda())
    torch.cuda.synchronize()
    assert_close(from_ts(yt), ref, rtol=2e-4, what=f"conv fwd {case}")


def test_conv_channel_range_launch(ops):
    """unet_conv_desc.cout_begin / cout_count: three launches over disjoint channel ranges of one 176-wide conv (with bias,
    residual and ReLU) reproduce the single launch bit for bit and touch nothing outside their range"""
    from unet_amd import _lib as L
    N, H, W, Cin, Cout = 2, 20, 36, 48, 176
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    xt, rt = to_ts(x), to_ts(r)
    wp = ops.pack_weights(w.cuda(), 0)
    full = empty_ts(N, H, W, Cout)
    d = ops._conv_desc(xt, wp, full, 3, 1, L.CONV_FWD, b.cuda(), rt, None, True, None, None)
    L.check(L.lib.unet_conv2d(L.C.byref(d), ops._stream()), "full")
    parts = empty_ts(N, H, W, Cout)
    for beg, cnt in ((0, 64), (64, 96), (160, 16)):
        d2 = ops._conv_desc(xt, wp, parts, 3, 1, L.CONV_FWD, b.cuda(), rt, None, True, None, None)
        d2.cout_begin, d2.cout_count = beg, cnt
        L.check(L.lib.unet_conv2d(L.C.byref(d2), ops._stream()), "part")
        torch.cuda.synchronize()
        got = from_ts(parts)
        assert torch.equal(got[:, beg:beg + cnt], from_ts(full)[:, beg:beg + cnt])
        assert bool((got[:, beg + cnt:] == 7.25).all())          # channels of later ranges still hold the fill value
    assert torch.equal(from_ts(parts), from_ts(full))
    ref = F.relu(F.conv2d(x, w, b, padding=1) + r)
    assert_close(from_ts(full), ref, rtol=2e-4, what="conv 176")
    d3 = ops._conv_desc(xt, wp, parts, 3, 1, L.CONV_FWD, None, None, None, False, None, None)
    d3.cout_begin, d3.cout_count = 8, 16                      # not a multiple of 16
    assert L.lib.unet_conv2d(L.C.byref(d3), ops._stream()) != 0


@pytest.mark.parametrize("Cin,Cout", [(100, 100), (96, 97), (36, 99), (100, 116), (20, 228)])
def test_conv_sliver_last_channels_on_mfma4x4(ops, Cin, Cout):
    """Output widths of 16 n + (1..4): in the 128 x 128 tile the last channels run on v_mfma_f32_4x4x1 (conv_igemm16_kernel `sliver`,
    filters from the sliver image behind the packed one).  Forward with bias + residual + ReLU into a channel slice, ragged pixel
    tiles; then the input-gradient form (residual + ReLU mask) of a conv whose INPUT width is such a count."""
    N, H, W = 2, 150, 170                      # 400+ blocks of 128 pixels: the planner keeps the 128 x 128 tile
    from unet_amd._lib import lib
    _knobs.set_knob("mfma_shape", -1)                # (large grids now run the fp32 form of the 256-pixel kernel, 7 channel tiles for 100 outputs: this
    try:                                       # test is about the sliver instantiation that smaller grids keep using)
        _sliver_case(ops, Cin, Cout, N, H, W)
    finally:
        _knobs.set_knob("mfma_shape", -2)


def _sliver_case(ops, Cin, Cout, N, H, W):
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    cs_out = (Cout + 3) // 4 * 4 + 8
    xt, rt = to_ts(x), to_ts(r, cs=cs_out + 4, co=4)
    yt = empty_ts(N, H, W, Cout, cs=cs_out, co=4)
    wp = ops.pack_weights(w.cuda(), 0)
    assert ops.conv2d_variant(xt, wp, yt, 3, 1) == 321280
    ops.conv2d(xt, wp, yt, 3, 1, bias=b.cuda(), res=rt, relu=True)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x, w, b, padding=1) + r)
    assert_close(from_ts(yt), ref, rtol=2e-4, what="sliver fwd")
    assert outside_untouched(yt)
    # column sums switch the sliver off: same launch, all channels on 16-wide tiles -- the results agree to summation order
    y2 = empty_ts(N, H, W, Cout)
    rows = ops.conv_colsum_rows(xt, wp, y2, 3, 1, 0)
    ops.conv2d(xt, wp, y2, 3, 1, bias=b.cuda(), res=rt, relu=True, colsum=torch.zeros(rows, Cout, device="cuda"))
    assert_close(from_ts(y2), from_ts(yt), rtol=2e-5, what="sliver vs tiles")
    assert torch.equal(from_ts(y2)[:, :Cout // 16 * 16], from_ts(yt)[:, :Cout // 16 * 16]), "the full tiles are the same program"
    # input gradient of a conv Cout -> Cin: produces Cout channels
    dy = torch.randn(N, Cin, H, W, generator=g)
    wd = torch.randn(Cin, Cout, 3, 3, generator=g) / (Cin * 9) ** 0.5
    m = torch.randn(N, Cout, H, W, generator=g)
    dxt = empty_ts(N, H, W, Cout)
    ops.conv2d_dgrad(to_ts(dy), ops.pack_weights(wd.cuda(), 1), dxt, 3, 1, res=rt, mask=to_ts(m))
    torch.cuda.synchronize()
    dref = (torch.nn.grad.conv2d_input((N, Cout, H, W), wd, dy, padding=1) + r) * (m > 0)
    assert_close(from_ts(dxt), dref, rtol=2e-4, what="sliver dgrad")


def test_conv_fwd_slices_relu_res_colsum(ops):
    """Channel-sliced input/output (concat elimination), bias+residual+ReLU epilogue, column sums."""
    N, H, W, Cin, Cout = 2, 24, 40, 36, 100
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    pre = F.conv2d(x, w, b, padding=1) + r
    ref = F.relu(pre)
    xt = to_ts(x, cs=64, co=8)
    rt = to_ts(r, cs=128, co=4)
    yt = empty_ts(N, H, W, Cout, cs=160, co=32)
    wp = ops.pack_weights(w.cuda(), 0)
    rows = ops.conv_colsum_rows(xt, wp, yt, 3, 1, 0)
    cs_ = torch.zeros(rows, Cout, device="cuda")
    cq_ = torch.zeros(rows, Cout, device="cuda")
    ops.conv2d(xt, wp, yt, 3, 1, bias=b.cuda(), res=rt, relu=True, colsum=cs_, colsumsq=cq_)
    torch.cuda.synchronize()
    assert_close(from_ts(yt), ref, rtol=2e-4, what="conv+res+relu")
    assert outside_untouched(yt)
    assert_close(cs_.sum(0).cpu(), ref.sum((0, 2, 3)), rtol=2e-4, atol=1e-2, what="colsum")
    assert_close(cq_.sum(0).cpu(), (ref * ref).sum((0, 2, 3)), rtol=2e-4, atol=1e-2, what="colsumsq")


SPLITK_CASES = [
    # N, H, W, Cin, Cout, ks: small output grids with a long reduction (deep stages at small batch: cfg1, predict at batch 1)
    (2, 8, 8, 512, 512, 3),          # xresnet18 layer 4 of a 256 x 256 tile, batch 2
    (1, 16, 16, 1024, 512, 3),       # middle_conv
    (2, 8, 8, 520, 200, 3),          # reduction tail (520 = 32 * 16 + 8), produced channels 128 + 72: two channel-range launches
    (1, 16, 16, 2048, 100, 1),       # 1x1, 128 chunks, a sliver-width output (the split launch must not use the sliver image)
    (3, 7, 9, 384, 36, 3),           # ragged pixel tile, 64-wide channel block
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_conv_splitk(ops, case):
    """Split-K launches (unet_conv_desc.splitk_ws): the planner cuts the reduction of a small-grid launch into contiguous chunk ranges;
    the slabs are added in split order and the fused epilogue (bias, residual, ReLU, mask) moves into the reduce kernel.  Against fp64
    torch: as accurate as the unsplit launch or better (shorter accumulation chains); bit-reproducible; forward and input gradient."""
    from unet_amd._lib import lib
    N, H, W, Cin, Cout, ks = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    m = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2) + r.double()) * (m > 0)
    xt, rt, mt = to_ts(x), to_ts(r, cs=Cout + 12, co=4), to_ts(m)
    wp = ops.pack_weights(w.cuda(), 0)
    outs, errs = [], []
    for on in (1, 0, 1):
        _knobs.set_knob("conv_splitk", on)
        try:
            yt = empty_ts(N, H, W, Cout, cs=Cout + 8 + (-Cout) % 4, co=8)
            var = ops.conv2d_variant(xt, wp, yt, ks)
            ops.conv2d(xt, wp, yt, ks, bias=b.cuda(), res=rt, mask=mt, relu=True)
            torch.cuda.synchronize()
        finally:
            _knobs.set_knob("conv_splitk", 1)
        assert outside_untouched(yt)
        got = from_ts(yt)
        outs.append(got)
        errs.append(((got.double() - ref).norm() / ref.norm()).item())
        if Cout <= 128:
            assert (var >= 2000000) == bool(on), (case, on, var)         # the planner splits exactly when allowed
    assert torch.equal(outs[0], outs[2])
    print(f"split-K {case}: rel L2 vs fp64 split {errs[0]:.2e} unsplit {errs[1]:.2e}")
    assert errs[0] <= 3e-7 and errs[0] <= 1.2 * errs[1] + 2e-8
    # input gradient of the same layer (reduction over Cout)
    dy = torch.randn(N, Cout, H, W, generator=g)
    refd = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy.double(), padding=ks // 2)
    dxt = empty_ts(N, H, W, Cin)
    ops.conv2d_dgrad(to_ts(dy), ops.pack_weights(w.cuda(), 1), dxt, ks, 1)
    torch.cuda.synchronize()
    e = ((from_ts(dxt).double() - refd).norm() / refd.norm()).item()
    assert e <= 6e-7, (case, e)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(ops, case):
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cout * ks * ks) ** 0.5
    pad = (ks - 1) // 2
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    dy = torch.randn(N, Cout, OH, OW, generator=g)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, stride=stride, padding=pad)
    dyt = to_ts(dy)
    dxt = empty_ts(N, H, W, Cin)
    wp = ops.pack_weights(w.cuda(), 1)
    ops.conv2d_dgrad(dyt, wp, dxt, ks, stride)
    torch.cuda.synchronize()
    assert_close(from_ts(dxt), ref, rtol=2e-4, what=f"conv dgrad {case}")


def test_conv_dgrad_mask_res(ops):
    N, H, W, Cin, Cout = 2, 20, 36, 96, 100
    g = torch.Generator().manual_seed(5)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cout * 9) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g)
    act = F.relu(torch.randn(N, Cin, H, W, generator=g))
    extra = torch.randn(N, Cin, H, W, generator=g)
    ref = (torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, padding=1) + extra) * (act > 0)
    dxt = empty_ts(N, H, W, Cin)
    ops.conv2d_dgrad(to_ts(dy), ops.pack_weights(w.cuda(), 1), dxt, 3, 1, res=to_ts(extra), mask=to_ts(act))
    torch.cuda.synchronize()
    assert_close(from_ts(dxt), ref, rtol=2e-4, what="dgrad+res+mask")

@pytest.mark.gpu
@pytest.mark.parametrize("K,Cout", [(5, 100), (2, 99), (8, 16), (7, 130)])
def test_conv1x1_small_reduction_kernel(ops, K, Cout):
    """conv1x1_smallk_kernel (the segmentation head's input gradient, 5 -> 100): variant id 9; forward form with bias + residual + ReLU
    into a channel slice, gradient form with residual + mask; against torch on the CPU"""
    N, H, W = 2, 37, 29
    g = torch.Generator().manual_seed(K * 100 + Cout)
    x = torch.randn(N, K, H, W, generator=g)
    w = torch.randn(Cout, K, 1, 1, generator=g)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    xt, rt = to_ts(x, cs=ops.rup4(K) + 8, co=4), to_ts(r, cs=ops.rup4(Cout) + 4, co=4)
    yt = empty_ts(N, H, W, Cout, cs=ops.rup4(Cout) + 12, co=8)
    wp = ops.pack_weights(w.cuda(), 0)
    assert ops.conv2d_variant(xt, wp, yt, 1, 1) == 9
    ops.conv2d(xt, wp, yt, 1, 1, bias=b.cuda(), res=rt, relu=True)
    torch.cuda.synchronize()
    assert_close(from_ts(yt), F.relu(F.conv2d(x, w, b) + r), rtol=1e-5, what="small-K fwd")
    assert outside_untouched(yt)
    # as an input gradient: reduction over the K "output" channels of a forward conv Cout -> K
    wf = torch.randn(K, Cout, 1, 1, generator=g)
    dy = torch.randn(N, K, H, W, generator=g)
    act = F.relu(torch.randn(N, Cout, H, W, generator=g))
    ref = (torch.nn.grad.conv2d_input((N, Cout, H, W), wf, dy) + r) * (act > 0)
    dxt = empty_ts(N, H, W, Cout)
    ops.conv2d_dgrad(to_ts(dy), ops.pack_weights(wf.cuda(), 1), dxt, 1, 1, res=to_ts(r), mask=to_ts(act))
    torch.cuda.synchronize()
    assert_close(from_ts(dxt), ref, rtol=1e-5, what="small-K dgrad")

@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout: conv1x1_head_kernel (variant id 11) -- the segmentation head and its neighbours
    (2, 37, 29, 100, 5),        # the headline head: six full chunks + a tail of 4 (ONE transposed MFMA step), ragged last wave
    (1, 64, 64, 96, 2),         # BASELINE configs[0]'s two classes, whole chunks only
    (2, 16, 16, 128, 16),       # the widest form
    (1, 20, 20, 36, 3),         # tail of 4 behind two chunks
    (1, 9, 7, 22, 1),           # regression head (one output), tail of 6: two transposed steps
    (3, 5, 5, 9, 7),            # narrowest reduction the kernel takes: a tail only (three steps)
])
def test_conv1x1_head_kernel(ops, case):
    """the streaming 1x1 kernel for <= 16 produced channels: bias (+ ReLU) into a channel slice; against torch on the CPU, and the SAME BITS as
    the implicit-GEMM kernel (unet_tuning.conv_head1x1 = 0): one accumulation chain per logit, so masks and mosaic checksums do not move"""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    xt = to_ts(x, cs=ops.rup4(Cin) + 8, co=4)
    wp = ops.pack_weights(w.cuda(), 0)
    for relu in (False, True):
        ref = F.conv2d(x, w, b)
        ref = F.relu(ref) if relu else ref
        outs = []
        for on in (1, 2, 0):                      # 2: the other pixel-tile count per trip
            with _knobs.tuning(conv_head1x1=on, conv_splitk=0):       # (a split reduction is another chain: the planner cuts 128 channels on a small grid)
                yt = empty_ts(N, H, W, Cout, cs=ops.rup4(Cout) + 12, co=8)
                assert (ops.conv2d_variant(xt, wp, yt, 1, 1) == 11) == bool(on)
                ops.conv2d(xt, wp, yt, 1, 1, bias=b.cuda(), relu=relu)
                torch.cuda.synchronize()
            assert outside_untouched(yt)
            outs.append(from_ts(yt))
        assert_close(outs[0], ref, rtol=1e-5, what=f"head 1x1 {case}")
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"head 1x1 {case}: not the bits of the implicit-GEMM kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, stride: conv3x3_smallcin_kernel (variant id 10) -- the stem's first conv and its neighbours
    (2, 64, 64, 4, 32, 2),      # the headline stem conv0 (RGBI tile)
    (1, 37, 29, 3, 32, 2),      # RGB, odd image edges (stride-2 output ceil(H / 2))
    (2, 18, 22, 7, 24, 1),      # stride 1, two input vectors, 24 outputs = 3 groups of 8 (85 pixels per block, last lane idle)
    (1, 33, 31, 8, 32, 1),      # widest form: 8 inputs (two vectors), 32 outputs
    (2, 512, 512, 4, 32, 2),    # two headline tiles: the four-pixels-per-thread form (>= 1024 workgroups), ragged last workgroup never
    (3, 16, 16, 1, 20, 2),      # one band; 20 outputs: the last group stores its lower 4 channels only
])
def test_conv3x3_small_cin_kernel(ops, case):
    """the direct 3x3 kernel for <= 8 input channels: bias + ReLU into a channel slice, against torch on the CPU and against the
    implicit-GEMM kernel on the same operands (unet_tuning.conv_smallcin = 0)"""
    N, H, W, Cin, Cout, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=1))
    OH, OW = ref.shape[2:]
    xt = to_ts(x, cs=ops.rup4(Cin) + 8, co=4)
    wp = ops.pack_weights(w.cuda(), 0)
    outs = []
    for on in (1, 0):
        with _knobs.tuning(conv_smallcin=on):
            yt = empty_ts(N, OH, OW, Cout, cs=ops.rup4(Cout) + 12, co=8)
            assert (ops.conv2d_variant(xt, wp, yt, 3, stride) == 10) == bool(on)
            ops.conv2d(xt, wp, yt, 3, stride, bias=b.cuda(), relu=True)
            torch.cuda.synchronize()
        assert outside_untouched(yt)
        outs.append(from_ts(yt))
    assert_close(outs[0], ref, rtol=1e-5, what=f"small-Cin 3x3 {case}")
    assert_close(outs[0], outs[1], rtol=2e-4, what=f"small-Cin 3x3 vs implicit GEMM {case}")
    # launches it must leave alone: a residual, column sums (BatchNorm statistics fused into the producer)
    yt = empty_ts(N, OH, OW, Cout)
    d = ops._conv_desc(xt, wp, yt, 3, stride, 0, None, to_ts(ref), None, False, None, None)
    from unet_amd._lib import lib
    import ctypes
    assert int(lib.unet_conv2d_variant(ctypes.byref(d))) != 10


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (Cin, Cout, H, W, N): the 256-pixel tile in its fp32 form (conv_bf16_t256_kernel<.., float>): no reduction tail, no 4-channel sliver
    (96, 96, 250, 270, 2),      # 6 channel tiles, ragged image edges
    (32, 128, 256, 256, 2),     # 8 tiles, two chunks
    (64, 72, 128, 128, 4),      # 5 tiles (shared third tile)
    (128, 232, 64, 64, 16),     # two launches: a full block + a narrow one of 7 tiles (232 = 128 + 104)
    (256, 128, 64, 64, 16),     # a deep stage at the fp32 threshold (256 blocks of 256 pixels)
    (64, 64, 64, 64, 16),       # 64-wide channel block
    (48, 32, 64, 64, 16),       # 32-wide block
    (64, 128, 32, 24, 64),      # 16 x 16 pixel patches (outputs 16..31 pixels wide), ragged second patch column
    (100, 100, 256, 256, 2),    # reduction tail of 4 channels (transposed chunk, one of four MFMA steps); 100 outputs = 6 tiles + the 4-channel SLIVER
    (96, 100, 250, 270, 2),     # the sliver behind full chunks only, ragged image edges (pixel tiles partly outside the image)
    (52, 228, 128, 128, 4),     # two launches: a full block + a 7-tile block (100 = 6 x 16 + 4) that ends in the sliver; tail of 4 channels
    (36, 100, 128, 128, 4),     # the sliver behind one full chunk + a tail of 4 channels
    (40, 112, 128, 128, 4),     # tail of 8 channels (two steps)
    (13, 96, 128, 128, 4),      # the tail is the only chunk
    (256, 256, 32, 32, 16),     # 64 pixel tiles: the planner narrows the channel block to 64 so that 256 workgroups run
    (64, 128, 32, 32, 16),      # 64 pixel tiles: narrowed to 32 channels
    (512, 512, 16, 16, 16),     # 16 x 16 images (16-pixel patches), 16 pixel tiles x 16 blocks of 32 channels instead of a split reduction
])
def test_conv_f32_t256_kernel(ops, case):
    """forward with bias + residual + ReLU into a channel slice and the input gradient with residual + mask on the fp32 form of the 256-pixel
    kernel (variant ...7 / ...6), against torch on the CPU and against the generic kernel (unet_tuning.f32_big_tile = 0) on the same operands.
    Output widths of 16 n + 1..4 in a 7-tile block run their last channels as a v_mfma_f32_4x4x1 sliver (conv_bf16_t256_kernel<7, 32, float, true>:
    four partial chains per output element, summed in a fixed order) -- in both directions for the 100 -> 100 case."""
    from unet_amd._lib import lib
    Cin, Cout, H, W, N = case
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x, w, b, padding=1) + r)
    xt, rt = to_ts(x, cs=ops.rup4(Cin) + 8, co=4), to_ts(r, cs=ops.rup4(Cout) + 4, co=4)
    wp = ops.pack_weights(w.cuda(), 0)
    outs = []
    for sw in (-2, -1):
        _knobs.set_knob("mfma_shape", sw)
        try:
            yt = empty_ts(N, H, W, Cout, cs=ops.rup4(Cout) + 12, co=8)
            if sw == -2:
                assert ops.conv2d_variant(xt, wp, yt, 3, 1) % 10 in (6, 7)
            ops.conv2d(xt, wp, yt, 3, 1, bias=b.cuda(), res=rt, relu=True)
            torch.cuda.synchronize()
            assert outside_untouched(yt)
            outs.append(from_ts(yt))
        finally:
            _knobs.set_knob("mfma_shape", -2)
    assert_close(outs[0], ref, rtol=2e-4, what="t256 f32 fwd")
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * ref.abs().max().item()          # same k-ordered chains, possibly another MFMA-internal order
    dy = torch.randn(N, Cout, H, W, generator=g)
    act = F.relu(torch.randn(N, Cin, H, W, generator=g))
    extra = torch.randn(N, Cin, H, W, generator=g)
    if True:
        dref = (torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, padding=1) + extra) * (act > 0)
        dxt = empty_ts(N, H, W, Cin)
        wpd = ops.pack_weights(w.cuda(), 1)
        ops.conv2d_dgrad(to_ts(dy), wpd, dxt, 3, 1, res=to_ts(extra), mask=to_ts(act))
        torch.cuda.synchronize()
        assert_close(from_ts(dxt), dref, rtol=2e-4, what="t256 f32 dgrad")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (Cin, Cout, H, W, N): conv1x1_gemm_kernel<float> (variant 8) -- 1x1 / stride 1, whole 16-channel chunks, >= 256 blocks of 128 pixels x 128 channels
    (96, 384, 64, 64, 8),       # the final PixelShuffle conv's shape family: 3 channel blocks
    (64, 256, 61, 67, 5),       # ragged pixel count (20435 = 159 x 128 + 83: the last pixel tile is partly outside)
    (256, 96, 128, 128, 2),     # a narrow block of 6 tiles dealt 3 + 3 to the two waves (the 384 -> 96 input gradient's shape family)
    (16, 200, 128, 128, 2),     # one reduction chunk; 128 + 72 channels = two launches of ops.conv2d (channel ranges)
    (48, 132, 128, 128, 3),     # three chunks (odd count: the second register set ends on a repeated load), last block 4 channels wide
])
@pytest.mark.parametrize("mode", [1, 2])
def test_conv1x1_gemm_kernel(ops, case, mode):
    with ops.tuning(conv1x1_gemm=mode):       # 1: operands global -> VGPR, 2: the pixel chunk staged once per workgroup through LDS (256-pixel tile)
        _conv1x1_gemm_case(ops, case)


def _conv1x1_gemm_case(ops, case):
    """forward form with bias + residual + ReLU into a channel slice and gradient form with residual + mask on the flat-pixel GEMM kernel,
    against torch on the CPU and against the implicit-GEMM kernel (UNET_CONV1X1_GEMM=0 is the process-wide switch; here: a grid below
    the kernel's 256-block threshold cannot be forced, so the cross-check is torch alone)"""
    Cin, Cout, H, W, N = case
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x, w, b) + r)
    xt, rt = to_ts(x, cs=ops.rup4(Cin) + 8, co=4), to_ts(r, cs=ops.rup4(Cout) + 4, co=4)
    wp = ops.pack_weights(w.cuda(), 0)
    yt = empty_ts(N, H, W, Cout, cs=ops.rup4(Cout) + 12, co=8)
    assert ops.conv2d_variant(xt, wp, yt, 1, 1) == 8
    ops.conv2d(xt, wp, yt, 1, 1, bias=b.cuda(), res=rt, relu=True)
    torch.cuda.synchronize()
    assert outside_untouched(yt)
    assert_close(from_ts(yt), ref, rtol=2e-4, what="1x1 gemm fwd")
    if Cout % 16 == 0:          # the gradient form reduces over Cout: whole chunks only
        dy = torch.randn(N, Cout, H, W, generator=g)
        act = F.relu(torch.randn(N, Cin, H, W, generator=g))
        extra = torch.randn(N, Cin, H, W, generator=g)
        dref = (torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy) + extra) * (act > 0)
        dxt = empty_ts(N, H, W, Cin)
        wpd = ops.pack_weights(w.cuda(), 1)
        dyt = to_ts(dy)
        assert ops.conv2d_variant(dyt, wpd, dxt, 1, 1, kind=1) == 8 or Cin < 128
        ops.conv2d_dgrad(dyt, wpd, dxt, 1, 1, res=to_ts(extra), mask=to_ts(act))
        torch.cuda.synchronize()
        assert_close(from_ts(dxt), dref, rtol=2e-4, what="1x1 gemm dgrad")


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("case", [(96, 96, 64, 64, 8), (64, 32, 61, 67, 8), (32, 16, 128, 128, 2)])
def test_conv1x1_shuffle_is_conv_relu_pixelshuffle(ops, case, bf16, mode):
    with ops.tuning(conv1x1_gemm=mode):       # (0: the default = LDS-staged form of the kernel, 1: its direct form)
        _conv1x1_shuffle_case(ops, case, bf16)


def _conv1x1_shuffle_case(ops, case, bf16):
    """unet_conv_desc.pixel_shuffle: ONE launch = PixelShuffle(2)(relu(conv1x1(x) + b)) written into a channel slice of the up-sampled buffer
    (mode-2 packed image: columns in pixel-shuffle order); its adjoint unet_shuffle_bwd_xmask takes the ReLU mask from that slice.  Against
    torch: F.pixel_shuffle(F.relu(F.conv2d(..)), 2) and its autograd gradient w.r.t. the conv output."""
    Cin, nf, H, W, N = case
    dt = torch.bfloat16 if bf16 else torch.float32
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf16 else (lambda t: t)
    g = torch.Generator().manual_seed(Cin * 100 + nf)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    w = rnd(torch.randn(4 * nf, Cin, 1, 1, generator=g) / Cin ** 0.5)
    b = torch.randn(4 * nf, generator=g)
    yc = F.conv2d(x.double(), w.double(), b.double()).requires_grad_(True)
    ref = F.pixel_shuffle(F.relu(yc), 2)
    v = 8 if bf16 else 4
    xb = torch.full((N, H, W, Cin + 2 * v), 7.25)
    xb[..., v:v + Cin] = x.permute(0, 2, 3, 1)
    xt = ops.TS(xb.to(dt).cuda(), v, Cin)
    Xt = ops.TS(torch.full((N, 2 * H, 2 * W, nf + 2 * v), 7.25, dtype=dt, device="cuda"), v, nf)
    assert ops.conv1x1_shuffle_applies(xt, Xt)
    wp = ops.pack_weights(w.cuda(), 2, dtype=dt)
    ops.conv1x1_shuffle(xt, wp, Xt, bias=b.cuda(), relu=True)
    torch.cuda.synchronize()
    got = Xt.view().float().permute(0, 3, 1, 2).cpu()
    tol = (2.0 ** -8 if bf16 else 2e-5) * ref.abs().max().item()
    assert (got.double() - ref.detach()).abs().max().item() <= tol
    assert bool((Xt.buf[..., :v] == 7.25).all()) and bool((Xt.buf[..., v + nf:] == 7.25).all())          # the neighbouring slices are untouched
    # unet_conv_desc.ps_tail: the same launch appends another NHWC slice (the network input of the final concat) behind the shuffled channels
    for tc in (4, 3, 8):
        tsrc = torch.full((N, 2 * H, 2 * W, v + 16), 3.5)
        tsrc[..., v:v + tc] = rnd(torch.randn(N, 2 * H, 2 * W, tc, generator=g))
        tsrc[..., v + tc:v + (tc + 3) // 4 * 4] = 0.0                       # pad lanes of the source slice are zeros
        tail = ops.TS(tsrc.to(dt).cuda(), v, tc)
        X2 = ops.TS(torch.full((N, 2 * H, 2 * W, nf + 2 * v + 16), 7.25, dtype=dt, device="cuda"), v, nf)
        at = v + nf + 4                                                     # a quad-aligned place behind the slice (with a gap)
        assert ops.conv1x1_shuffle_tail_ok(X2, tail, at) and not ops.conv1x1_shuffle_tail_ok(X2, tail, at + 2)
        ops.conv1x1_shuffle(xt, wp, X2, bias=b.cuda(), relu=True, tail=tail, tail_at=at)
        torch.cuda.synchronize()
        tq = (tc + 3) // 4 * 4
        assert torch.equal(X2.view(), Xt.view()), "the shuffled channels changed with a tail"
        assert torch.equal(X2.buf[..., at:at + tq], tail.buf[..., v:v + tq]), "tail channels"
        rest = torch.ones(X2.buf.shape[-1], dtype=torch.bool)
        rest[v:v + nf] = False
        rest[at:at + tq] = False
        assert bool((X2.buf[..., rest] == 7.25).all()), "wrote outside the slice and the tail"
    # adjoint: dL/d(yc) from dL/dX, masked by the ReLU -- the mask read from X itself
    dX = rnd(torch.randn(N, nf, 2 * H, 2 * W, generator=g))
    ref.backward(dX.double())
    dXt = ops.TS(dX.permute(0, 2, 3, 1).contiguous().to(dt).cuda(), 0, nf)
    dyc = ops.TS(torch.empty((N, H, W, 4 * nf), dtype=dt, device="cuda"), 0, 4 * nf)
    ops.shuffle_bwd_xmask(dXt, Xt, dyc)
    torch.cuda.synchronize()
    gy = dyc.view().float().permute(0, 3, 1, 2).cpu().double()
    # (a pre-activation that rounds to exactly 0 in bf16 but is positive in fp64 is masked off here: compare where the stored activation decides)
    decided = (F.pixel_unshuffle(got.double(), 2) > 0) == (yc.detach() > 0)
    assert decided.float().mean().item() > 0.999
    assert ((gy - yc.grad).abs() * decided).max().item() == 0.0
    # a geometry the GEMM kernel does not take is refused, not silently planned elsewhere
    small = ops.TS(torch.zeros((1, 8, 8, Cin), dtype=dt, device="cuda"), 0, Cin)
    assert not ops.conv1x1_shuffle_applies(small, ops.TS(torch.zeros((1, 16, 16, nf), dtype=dt, device="cuda"), 0, nf))


WGRAD_CASES = [
    (2, 16, 16, 32, 32, 3, 1),
    (2, 40, 48, 100, 100, 3, 1),
    (1, 8, 8, 256, 128, 3, 1),
    (2, 13, 13, 64, 64, 3, 1),
    (2, 32, 32, 4, 32, 3, 2),
    (1, 16, 16, 3, 32, 3, 2),
    (2, 26, 26, 64, 128, 3, 2),
    (1, 25, 25, 32, 64, 3, 2),
    (2, 16, 16, 128, 256, 1, 1),
    (2, 32, 32, 100, 5, 1, 1),        # head: FMA kernel for <= 16 output channels
    (1, 17, 19, 8, 16, 1, 1),
    (3, 9, 7, 36, 3, 1, 1),
    (8, 128, 128, 100, 10, 1, 1),     # many workgroups, 10 classes
    (4, 64, 64, 64, 64, 3, 1),        # several split-K partials
    # narrow-output kernel (taps flattened into the column dimension): 80 < Cout <= 112, width >= 32
    (1, 33, 64, 192, 96, 3, 1),       # two input-channel chunks of 96, 6 output tiles x 7 column tiles per wave
    (2, 32, 40, 96, 96, 3, 1),        # ragged last tile row (40 = 32 + 8)
    (1, 35, 37, 100, 112, 3, 1),      # full 7 output tiles
    (2, 34, 32, 36, 100, 3, 1),       # short chunk: 9 * 36 = 324 columns, most column blocks idle
    (1, 32, 33, 230, 84, 3, 1),       # three chunks of 80/80/70 channels (chunk width not a divisor of Cin)
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(ops, case):
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 2)
    pad = (ks - 1) // 2
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, OH, OW, generator=g)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, ks, ks), dy, stride=stride, padding=pad)
    xt, dyt = to_ts(x), to_ts(dy)
    ws = torch.empty(ops.wgrad_workspace(xt, dyt, ks, stride, with_bias=True), device="cuda")
    dw = torch.full((Cout, Cin, ks, ks), 3.0, device="cuda")
    db = torch.full((Cout,), 3.0, device="cuda")
    ops.conv2d_wgrad(xt, dyt, dw, ks, stride, ws, dbias=db)
    torch.cuda.synchronize()
    assert_close(dw.cpu(), ref, rtol=3e-4, atol=1e-4, what=f"wgrad {case}")
    assert_close(db.cpu(), dy.sum((0, 2, 3)), rtol=3e-4, atol=1e-4, what=f"dbias {case}")
    ops.conv2d_wgrad(xt, dyt, dw, ks, stride, ws, accumulate=True)
    torch.cuda.synchronize()
    assert_close(dw.cpu(), 2 * ref, rtol=3e-4, atol=2e-4, what=f"wgrad accumulate {case}")


@pytest.mark.parametrize("mode", [2, 0])
def test_wgrad_1x1_kernels_agree(ops, mode):
    """1x1 weight gradients on the 128 x 128-blocked GEMM kernel (unet_tuning.wgrad_1x1 = 2: forced; the planner keeps it for launches of more
    than 64 channels on both sides and >= 3 GFLOP) and on the 64 x 64-blocked general kernel (0), against torch: ragged channel counts on
    both sides of the block sizes, SelfAttention's 48-wide operands, bias gradient, accumulate"""
    for case in [(2, 16, 16, 128, 256), (1, 24, 40, 48, 300), (1, 24, 40, 300, 48), (2, 17, 19, 130, 70), (1, 64, 64, 96, 384)]:
        N, H, W, Cin, Cout = case
        g = torch.Generator().manual_seed(sum(case) + mode)
        x = torch.randn(N, Cin, H, W, generator=g)
        dy = torch.randn(N, Cout, H, W, generator=g)
        ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 1, 1), dy)
        xt, dyt = to_ts(x), to_ts(dy)
        with ops.tuning(wgrad_1x1=mode):
            ws = torch.empty(ops.wgrad_workspace(xt, dyt, 1, 1, with_bias=True), device="cuda")
            dw = torch.full((Cout, Cin, 1, 1), 3.0, device="cuda")
            db = torch.full((Cout,), 3.0, device="cuda")
            ops.conv2d_wgrad(xt, dyt, dw, 1, 1, ws, dbias=db)
            torch.cuda.synchronize()
            assert_close(dw.cpu(), ref, rtol=3e-4, atol=1e-4, what=f"1x1 wgrad mode {mode} {case}")
            assert_close(db.cpu(), dy.sum((0, 2, 3)), rtol=3e-4, atol=1e-4, what=f"1x1 dbias mode {mode} {case}")
            ops.conv2d_wgrad(xt, dyt, dw, 1, 1, ws, accumulate=True)
            torch.cuda.synchronize()
            assert_close(dw.cpu(), 2 * ref, rtol=3e-4, atol=2e-4, what=f"1x1 wgrad accumulate mode {mode} {case}")


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 32, 9, 7), (1, 512, 4, 4), (2, 256, 40, 40), (1, 512, 2, 2), (16, 64, 128, 128)])
def test_batchnorm_train_fwd_bwd(ops, shape):
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    res = torch.randn(shape, generator=g)
    gamma = (torch.rand(Cc, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(Cc, generator=g).requires_grad_(True)
    rm0, rv0 = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    rm, rv = rm0.clone(), rv0.clone()
    bn = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    out = F.relu(bn + res)
    dout = torch.randn(shape, generator=g)

    P = N * H * W
    xt = to_ts(x.detach())
    rows = ops.bn_stats_rows(P)
    part = torch.empty(2 * rows * Cc, device="cuda")
    ops.bn_stats(xt, part)
    dev = lambda t: t.detach().clone().cuda()
    scale, shift, smean, sinv = (torch.empty(Cc, device="cuda") for _ in range(4))
    rmd, rvd = dev(rm0), dev(rv0)
    tracked = torch.full((), 7, dtype=torch.int64, device="cuda")
    ops.bn_finalize(part, part[rows * Cc:], rows, P, Cc, dev(gamma), dev(beta), rmd, rvd, 0.1, 1e-5, scale, shift, smean, sinv, tracked)
    yt = empty_ts(N, H, W, Cc)
    ops.affine_act(xt, yt, scale, shift, x2=to_ts(res), relu=True)
    torch.cuda.synchronize()
    assert_close(from_ts(yt), out.detach(), rtol=1e-5, atol=1e-5, what="bn fwd")
    assert_close(rmd.cpu(), rm, rtol=1e-5, atol=1e-6, what="running_mean")
    assert_close(rvd.cpu(), rv, rtol=1e-5, atol=1e-6, what="running_var")
    assert int(tracked.item()) == 8          # BatchNorm2d.num_batches_tracked += 1 inside the finalize kernel

    # backward: g = dout * (out > 0); dx, dgamma, dbeta; residual branch gets g.  The reference uses the
    # ReLU mask of the HIP forward output so that a sign flip of a ~1e-8 pre-activation cannot fail the test.
    g_ref = dout * (from_ts(yt) > 0)
    bn.backward(g_ref)
    doutt = to_ts(dout)
    bpart = torch.empty(2 * rows * Cc, device="cuda")
    ops.bn_bwd_reduce(doutt, yt, xt, smean, sinv, bpart)
    dgamma, dbeta, c1, c2 = (torch.empty(Cc, device="cuda") for _ in range(4))
    ops.bn_bwd_finalize(bpart, rows, P, Cc, dgamma, dbeta, c1, c2)
    dxt = empty_ts(N, H, W, Cc)
    gt = empty_ts(N, H, W, Cc)
    ops.bn_bwd_apply(doutt, yt, xt, smean, sinv, dev(gamma), c1, c2, dxt, gout=gt)
    torch.cuda.synchronize()
    assert_close(from_ts(dxt), x.grad, rtol=2e-4, atol=1e-5, what="bn dx")
    assert_close(dgamma.cpu(), gamma.grad, rtol=2e-4, atol=1e-4, what="dgamma")
    assert_close(dbeta.cpu(), beta.grad, rtol=2e-4, atol=1e-4, what="dbeta")
    assert torch.equal(from_ts(gt), g_ref)


def test_batchnorm_eval_coeffs(ops):
    Cc = 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, Cc, 8, 8, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    ref = F.batch_norm(x, rm, rv, gamma, beta, training=False, eps=1e-5)
    scale, shift = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    ops.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5, scale, shift)
    yt = empty_ts(2, 8, 8, Cc)
    ops.affine_act(to_ts(x), yt, scale, shift)
    torch.cuda.synchronize()
    assert_close(from_ts(yt), ref, rtol=1e-5, atol=1e-5, what="bn eval")


@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (1, 32, 25, 27), (2, 8, 7, 7)])
def test_maxpool(ops, shape):
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(4)
    x = F.relu(torch.randn(shape, generator=g)).requires_grad_(True)   # many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    OH, OW = y.shape[-2:]
    yt = empty_ts(N, OH, OW, Cc)
    idx = torch.empty(N * OH * OW * Cc, dtype=torch.uint8, device="cuda")
    ops.maxpool(to_ts(x.detach()), yt, idx)
    dxt = empty_ts(N, H, W, Cc)
    ops.maxpool_bwd(to_ts(dy), idx, dxt)
    torch.cuda.synchronize()
    assert torch.equal(from_ts(yt), y.detach())
    assert_close(from_ts(dxt), x.grad, rtol=1e-6, atol=1e-6, what="maxpool bwd")


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 128, 25, 13)])
def test_avgpool_ceil(ops, shape):
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(shape, generator=g).requires_grad_(True)
    y = F.avg_pool2d(x, 2, ceil_mode=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    OH, OW = y.shape[-2:]
    yt = empty_ts(N, OH, OW, Cc)
    ops.avgpool(to_ts(x.detach()), yt)
    dxt = empty_ts(N, H, W, Cc)
    ops.avgpool_bwd(to_ts(dy), dxt)
    torch.cuda.synchronize()
    assert_close(from_ts(yt), y.detach(), rtol=1e-6, atol=1e-6, what="avgpool")
    assert_close(from_ts(dxt), x.grad, rtol=1e-6, atol=1e-6, what="avgpool bwd")


@pytest.mark.parametrize("blur", [True, False])
@pytest.mark.parametrize("shape", [(2, 24, 8, 8), (1, 96, 5, 7)])
def test_shuffle_blur(ops, shape, blur):
    N, Cu, h, w = shape
    g = torch.Generator().manual_seed(8)
    pre = torch.randn(N, 4 * Cu, h, w, generator=g).requires_grad_(True)
    yc = F.relu(pre)
    up = F.pixel_shuffle(yc, 2)
    if blur:
        up = F.avg_pool2d(F.pad(up, (1, 0, 1, 0), mode="replicate"), 2, stride=1)
    dX = torch.randn(up.shape, generator=g)
    up.backward(dX)
    yct = to_ts(yc.detach())
    Xt = empty_ts(N, 2 * h, 2 * w, Cu, cs=Cu + 12, co=4)
    ops.shuffle_blur(yct, Xt, blur)
    dyct = empty_ts(N, h, w, 4 * Cu)
    ops.shuffle_blur_bwd(to_ts(dX, cs=Cu + 8, co=8), yct, dyct, blur)
    torch.cuda.synchronize()
    assert_close(from_ts(Xt), up.detach(), rtol=1e-6, atol=1e-6, what="shuffle_blur")
    assert outside_untouched(Xt)
    assert_close(from_ts(dyct), pre.grad, rtol=1e-5, atol=1e-6, what="shuffle_blur bwd")


def test_resize_nearest(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 26, 26, generator=g).requires_grad_(True)
    y = F.interpolate(x, (25, 25), mode="nearest")
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yt = empty_ts(2, 25, 25, 32)
    ops.resize_nearest(to_ts(x.detach()), yt)
    dxt = empty_ts(2, 26, 26, 32)
    ops.resize_nearest_bwd(to_ts(dy), dxt)
    torch.cuda.synchronize()
    assert torch.equal(from_ts(yt), y.detach())
    assert_close(from_ts(dxt), x.grad, rtol=1e-6, atol=1e-6, what="resize bwd")
    x2 = torch.randn(1, 8, 50, 50, generator=g)
    y2t = empty_ts(1, 52, 52, 8)
    ops.resize_nearest(to_ts(x2), y2t)
    torch.cuda.synchronize()
    assert torch.equal(from_ts(y2t), F.interpolate(x2, (52, 52), mode="nearest"))


def test_layout_and_slices(ops):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 3, 9, 11, generator=g)
    yt = empty_ts(2, 9, 11, 3, cs=8, co=4, fill=0.0)
    ops.nchw_to_nhwc(x.cuda(), yt)
    back = torch.empty(2, 3, 9, 11, device="cuda")
    ops.nhwc_to_nchw(yt, back)
    torch.cuda.synchronize()
    assert torch.equal(from_ts(yt), x) and torch.equal(back.cpu(), x)
    a = torch.randn(2, 20, 5, 5, generator=g)
    b = torch.randn(2, 20, 5, 5, generator=g)
    at = to_ts(a, cs=32, co=4)
    bt = to_ts(b)
    ops.copy_slice(at, bt, accumulate=True)
    mt = empty_ts(2, 5, 5, 20)
    ops.relu_mask(at, bt, mt)
    ws = torch.empty(ops.colsum_workspace(50, 20), device="cuda")
    cs_ = torch.empty(20, device="cuda")
    ops.colsum(at, cs_, ws)
    torch.cuda.synchronize()
    assert_close(from_ts(bt), a + b, rtol=1e-6, atol=1e-6, what="copy accumulate")
    assert torch.equal(from_ts(mt), a * ((a + b) > 0))
    assert_close(cs_.cpu(), a.sum((0, 2, 3)), rtol=1e-5, atol=1e-5, what="colsum")


@pytest.mark.parametrize("n_cls,weighted", [(5, True), (5, False), (2, True), (10, False)])
def test_cross_entropy(ops, n_cls, weighted):
    g = torch.Generator().manual_seed(12)
    N, H, W = 2, 19, 23
    z = (torch.randn(N, n_cls, H, W, generator=g) * 3).requires_grad_(True)
    y = torch.randint(0, n_cls, (N, H, W), generator=g)
    wt = (torch.rand(n_cls, generator=g) + 0.2) if weighted else None
    zl = z.permute(0, 2, 3, 1).reshape(-1, n_cls)
    loss = F.cross_entropy(zl, y.reshape(-1), weight=wt)
    loss.backward()
    zt = to_ts(z.detach(), cs=ops.rup4(n_cls) + 4, co=4)
    P = N * H * W
    ws = torch.empty(ops.ce_workspace(P), device="cuda")
    lo, den = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    wd = None if wt is None else wt.cuda()
    ops.ce_fwd(zt, y.cuda(), wd, lo, den, ws)
    dzt = empty_ts(N, H, W, n_cls, fill=0.0)
    ops.ce_bwd(zt, y.cuda(), wd, den, 1.0, dzt)
    probs = torch.empty(N, n_cls, H, W, device="cuda")
    am = torch.empty(N, H, W, dtype=torch.int64, device="cuda")
    ops.softmax_argmax(zt, probs, am)
    torch.cuda.synchronize()
    assert abs(lo.item() - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))
    assert_close(from_ts(dzt), z.grad, rtol=1e-4, atol=1e-9, what="ce bwd")
    ref_p = F.softmax(z.detach(), dim=1)
    assert_close(probs.cpu(), ref_p, rtol=1e-5, atol=1e-6, what="softmax")
    assert torch.equal(am.cpu(), ref_p.argmax(dim=1))


def test_adam_step_matches_fastai_restatement(ops):
    from oracle.unet_oracle import FastaiAdam
    g = torch.Generator().manual_seed(13)
    n = 10000
    p0 = torch.randn(n, generator=g)
    ps = [torch.nn.Parameter(p0[:3000].clone()), torch.nn.Parameter(p0[3000:7000].clone()), torch.nn.Parameter(p0[7000:].clone())]
    lrs = [1e-3, 3e-3, 1e-2]
    opt = FastaiAdam([[ps[0]], [ps[1]], [ps[2]]], lrs, no_wd=[ps[1]])
    code = torch.empty(n, dtype=torch.uint8)
    code[:3000] = 0 | 4
    code[3000:7000] = 1
    code[7000:] = 2 | 4
    pd = p0.clone().cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    coded = code.cuda()
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        for q, sl in zip(ps, (slice(0, 3000), slice(3000, 7000), slice(7000, n))):
            q.grad = gr[sl].clone()
        opt.mom = 0.95 - 0.02 * step
        opt.step()
        ops.adam_step(pd, gr.cuda(), m, v, coded, lrs, opt.mom, 0.99, 1e-5, 0.01, step)
    torch.cuda.synchronize()
    ref = torch.cat([q.detach() for q in ps])
    assert_close(pd.cpu(), ref, rtol=1e-6, atol=1e-6, what="adam")


def test_mosaic(ops):
    g = torch.Generator().manual_seed(14)
    Cc, MH, MW = 3, 20, 24
    mosaic = torch.zeros(Cc, MH, MW, device="cuda")
    count = torch.zeros(MH, MW, dtype=torch.int32, device="cuda")
    ref = torch.zeros(Cc, MH, MW)
    cnt = torch.zeros(MH, MW)
    for (y0, x0) in [(0, 0), (4, 8), (8, 12), (12, 12)]:
        pr = torch.rand(Cc, 8, 12, generator=g)
        ops.mosaic_accumulate(pr.cuda(), mosaic, count, y0, x0)
        ref[:, y0:y0 + 8, x0:x0 + 12] += pr
        cnt[y0:y0 + 8, x0:x0 + 12] += 1
    am = torch.empty(MH, MW, dtype=torch.uint8, device="cuda")
    ops.mosaic_finalize(mosaic, count, am)
    torch.cuda.synchronize()
    refm = torch.where(cnt > 0, ref / cnt.clamp(min=1), ref)
    assert_close(mosaic.cpu(), refm, rtol=1e-6, atol=1e-6, what="mosaic")
    assert torch.equal(am.cpu().long(), refm.argmax(0))


def test_bad_arguments_fail_loudly(ops):
    from unet_amd._lib import UnetHipError
    x = empty_ts(1, 8, 8, 16)
    y = empty_ts(1, 9, 9, 16)
    wp = torch.zeros(9 * 128 * 16, device="cuda")
    with pytest.raises(UnetHipError):
        ops.conv2d(x, wp, y, 3, 1)          # inconsistent output dims
    with pytest.raises(UnetHipError):
        ops.conv2d(x, wp, empty_ts(1, 8, 8, 16), 5, 1)   # unsupported kernel size


@pytest.mark.parametrize("shape", [16, 32])
def test_conv_mfma_shapes_agree(ops, shape):
    """both MFMA instruction shapes (16x16x4 with tile skipping, 32x32x2) give the reference result on the awkward widths"""
    from unet_amd._lib import lib
    assert _knobs.set_knob("mfma_shape", shape) == 0
    try:
        for case in [(2, 40, 48, 100, 100, 3, 1), (1, 16, 16, 192, 96, 3, 1), (1, 32, 32, 99, 2, 1, 1), (2, 26, 26, 64, 128, 3, 2),
                     (1, 24, 24, 36, 100, 3, 1), (1, 16, 16, 20, 52, 3, 1)]:
            N, H, W, Cin, Cout, ks, stride = case
            g = torch.Generator().manual_seed(sum(case))
            x = torch.randn(N, Cin, H, W, generator=g)
            w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
            b = torch.randn(Cout, generator=g)
            ref = _conv_ref(x, w, b, ks, stride)
            yt = empty_ts(N, ref.shape[2], ref.shape[3], Cout)
            rows = ops.conv_colsum_rows(to_ts(x), ops.pack_weights(w.cuda(), 0), yt, ks, stride, 0)
            cs_ = torch.zeros(rows, Cout, device="cuda")
            ops.conv2d(to_ts(x), ops.pack_weights(w.cuda(), 0), yt, ks, stride, bias=b.cuda(), colsum=cs_)
            dy = torch.randn(ref.shape, generator=g)
            dxt = empty_ts(N, H, W, Cin)
            ops.conv2d_dgrad(to_ts(dy), ops.pack_weights(w.cuda(), 1), dxt, ks, stride)
            torch.cuda.synchronize()
            assert_close(from_ts(yt), ref, rtol=2e-4, what=f"mf{shape} fwd {case}")
            assert_close(cs_.sum(0).cpu(), ref.sum((0, 2, 3)), rtol=2e-4, atol=1e-2, what=f"mf{shape} colsum {case}")
            assert_close(from_ts(dxt), torch.nn.grad.conv2d_input(x.shape, w, dy, stride=stride, padding=(ks - 1) // 2), rtol=2e-4,
                         what=f"mf{shape} dgrad {case}")
    finally:
        _knobs.set_knob("mfma_shape", 16)


def test_wgrad_mfma_shapes_agree(ops):
    from unet_amd._lib import lib
    for shape in (32, 16):
        assert _knobs.set_knob("wgrad_mfma_shape", shape) == 0
        try:
            for case in [(2, 40, 48, 100, 100, 3, 1), (1, 16, 16, 192, 96, 3, 1), (2, 26, 26, 64, 128, 3, 2), (2, 16, 16, 100, 5, 1, 1),
                         (1, 24, 24, 36, 52, 3, 1)]:
                N, H, W, Cin, Cout, ks, stride = case
                g = torch.Generator().manual_seed(sum(case))
                pad = (ks - 1) // 2
                OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
                x = torch.randn(N, Cin, H, W, generator=g)
                dy = torch.randn(N, Cout, OH, OW, generator=g)
                ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, ks, ks), dy, stride=stride, padding=pad)
                xt, dyt = to_ts(x), to_ts(dy)
                ws = torch.empty(ops.wgrad_workspace(xt, dyt, ks, stride), device="cuda")
                dw = torch.empty((Cout, Cin, ks, ks), device="cuda")
                ops.conv2d_wgrad(xt, dyt, dw, ks, stride, ws)
                torch.cuda.synchronize()
                assert_close(dw.cpu(), ref, rtol=3e-4, atol=1e-4, what=f"wgrad mf{shape} {case}")
        finally:
            _knobs.set_knob("wgrad_mfma_shape", 32)


def test_row_softmax_and_strided_pack(ops):
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(2, 300, 3, 5, generator=g) * 3).requires_grad_(True)     # [N, C, H, W]: softmax over C per pixel
    y = torch.softmax(x, dim=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xt = to_ts(x.detach())
    yt = empty_ts(2, 3, 5, 300)
    ops.row_softmax(xt, yt)
    dxt = to_ts(dy)
    ops.row_softmax_bwd(yt, dxt, dxt)          # in place
    torch.cuda.synchronize()
    assert_close(from_ts(yt), y.detach(), rtol=1e-5, atol=1e-7, what="row softmax")
    assert_close(from_ts(dxt), x.grad, rtol=1e-4, atol=1e-7, what="row softmax bwd")
    # strided pack: a [O, R] matrix stored transposed inside a wider activation buffer behaves like a 1x1 filter
    O_, R = 52, 36
    act = torch.randn(R, 80, generator=g)                 # rows r, O values at columns 8..8+O
    w = act[:, 8:8 + O_].t().contiguous()                 # [O, R]
    xin = torch.randn(1, R, 6, 7, generator=g)
    ref = F.conv2d(xin, w.view(O_, R, 1, 1))
    actd = act.cuda()
    wp = torch.empty(ops.lib.unet_pack_weights_size(O_, R, 1, 0), device="cuda")
    ops.pack_weights_strided(actd.data_ptr() + 8 * 4, 1, 80, O_, R, wp)
    out = empty_ts(1, 6, 7, O_)
    ops.conv2d(to_ts(xin), wp, out, 1)
    torch.cuda.synchronize()
    assert_close(from_ts(out), ref, rtol=2e-4, what="strided pack conv")
    # an output width of 16 n + 1..4 with ONE image for the whole batch (wp_img_stride 0) at a size where the planner picks the 128 x 128
    # tile: that launch reads the 4x4x1 sliver block behind the image, which the strided packer must write too
    O_, R = 100, 40
    act = torch.randn(R, 120, generator=g)
    w = act[:, 12:12 + O_].t().contiguous()
    xin = torch.randn(4, R, 128, 128, generator=g)
    ref = F.conv2d(xin, w.view(O_, R, 1, 1))
    actd = act.cuda()
    wp = torch.full((ops.lib.unet_pack_weights_size(O_, R, 1, 0),), float("nan"), device="cuda")
    ops.pack_weights_strided(actd.data_ptr() + 12 * 4, 1, 120, O_, R, wp)
    assert not bool(torch.isnan(wp).any())                # every element of the image, sliver included, is written
    wp_ref = ops.pack_weights(w.view(O_, R, 1, 1).contiguous().cuda(), 0)
    assert torch.equal(wp, wp_ref)                        # "the same packed image as mode 0 with ks = 1"
    out = empty_ts(4, 128, 128, O_)
    ops.conv2d(to_ts(xin), wp, out, 1)
    torch.cuda.synchronize()
    assert_close(from_ts(out), ref, rtol=2e-4, what="strided pack conv, sliver width")


@pytest.mark.parametrize("narrow", [1, 2, 0])
def test_wgrad_narrow_kernel(ops, narrow):
    """narrow-output weight gradient (80 < Cout <= 112, 3x3 s1, W >= 32; default on: 97..100 output channels with the last 1..4 on the
    v_mfma_f32_4x4x1 sliver), the same with seven full 16-row tiles (2) and its 64x64-tiled fallback (0) on the same cases against torch,
    incl. bias gradient, ragged tiles and input-channel chunking"""
    from unet_amd._lib import lib
    _knobs.set_knob("wgrad_narrow", narrow)
    try:
        _narrow_cases(ops)
    finally:
        _knobs.set_knob("wgrad_narrow", 1)


def _narrow_cases(ops):
    for case in [(2, 40, 48, 100, 100), (1, 34, 64, 192, 96), (2, 32, 32, 96, 96), (1, 37, 45, 36, 100), (1, 32, 40, 250, 81),
                 (3, 5, 33, 100, 100), (1, 33, 40, 40, 98), (2, 32, 32, 100, 97), (1, 32, 64, 57, 104)]:
        N, H, W, Cin, Cout = case
        g = torch.Generator().manual_seed(sum(case))
        x = torch.randn(N, Cin, H, W, generator=g)
        dy = torch.randn(N, Cout, H, W, generator=g)
        ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
        xt, dyt = to_ts(x), to_ts(dy)
        ws = torch.empty(ops.wgrad_workspace(xt, dyt, 3, 1), device="cuda")
        dw = torch.empty((Cout, Cin, 3, 3), device="cuda")
        db = torch.empty(Cout, device="cuda")
        ops.conv2d_wgrad(xt, dyt, dw, 3, 1, ws, dbias=db)
        torch.cuda.synchronize()
        assert_close(dw.cpu(), ref, rtol=3e-4, atol=1e-4, what=f"narrow wgrad {case}")
        assert_close(db.cpu(), dy.sum((0, 2, 3)), rtol=3e-4, atol=1e-4, what=f"narrow dbias {case}")


@pytest.mark.parametrize("kind", ["mse", "l1", "smoothl1"])
def test_regression_loss_kernels(ops, kind):
    """unet_regloss_fwd / _bwd against torch (MSELoss / L1Loss / SmoothL1Loss(beta=0.5), mean reduction) on a channel slice"""
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 13, 17
    z = torch.randn(N, 1, H, W, generator=g, requires_grad=True)
    t = torch.randn(N, H, W, generator=g)
    t.view(-1)[:5] = z.detach().view(-1)[:5]                      # exact zeros of the residual (sign(0) = 0)
    fn = {"mse": F.mse_loss, "l1": F.l1_loss, "smoothl1": lambda a, b: F.smooth_l1_loss(a, b, beta=0.5)}[kind]
    ref = fn(z.reshape(-1), t.reshape(-1))
    ref.backward()
    zt = to_ts(z.detach(), cs=8, co=4)
    dz = empty_ts(N, H, W, 1, cs=12, co=8)
    loss = torch.zeros(1, device="cuda")
    ws = torch.empty(ops.ce_workspace(zt.P), device="cuda")
    ops.regloss_fwd(zt, t.cuda().contiguous(), kind, 0.5, loss, ws)
    ops.regloss_bwd(zt, t.cuda().contiguous(), kind, 0.5, 0.5, dz)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert_close(from_ts(dz), 0.5 * z.grad, rtol=1e-5, atol=1e-8, what=f"regloss bwd {kind}")
    assert outside_untouched(dz)
