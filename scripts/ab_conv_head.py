"""The segmentation head (100 -> 5, 1x1) alone on the GPU: conv1x1_head_kernel against the implicit-GEMM kernels (unet_tuning.conv_head1x1 = 0),
both storage types, batch 16 and 1 of 512^2 tiles; and the layout conversion of the input tile.  usage: python scripts/ab_conv_head.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for dt in (torch.float32, torch.bfloat16):
    es = 2 if dt == torch.bfloat16 else 4
    for N, Cin, Cout, H in [(16, 100, 5, 512), (1, 100, 5, 512), (16, 100, 2, 512), (16, 96, 5, 256)]:
        x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
        x.buf[..., Cin:] = 0
        w = torch.randn((Cout, Cin, 1, 1), device="cuda", generator=g) / Cin ** 0.5
        b = torch.randn(Cout, device="cuda", generator=g)
        wf = ops.pack_weights(w, 0, dtype=dt)
        res = []
        for on in (1, 2, 0):
            with ops.tuning(conv_head1x1=on):
                y = TS(torch.empty((N, H, H, ops.rup4(Cout)), device="cuda", dtype=torch.float32), 0, Cout)
                var = ops.conv2d_variant(x, wf, y, 1, 1)
                t = timeit(lambda: ops.conv2d(x, wf, y, 1, 1, bias=b))
                res.append((var, t, y.buf[..., :Cout].clone()))
        by = N * H * H * (es * ops.rupv(Cin, dt) + 4 * ops.rup4(Cout))
        print(f"{str(dt)[6:]:9s} N{N:2d} {Cin}->{Cout} @{H}: head {res[0][1]:7.1f} us ({by / res[0][1] / 1e6:5.2f} TB/s, variant {res[0][0]})  other tile count {res[1][1]:7.1f}  "
              f"igemm {res[2][1]:7.1f} us (variant {res[2][0]})  same bits: {all(torch.equal(res[0][2], r[2]) for r in res[1:])}", flush=True)
for dt in (torch.float32, torch.bfloat16):
    x = torch.rand(16, 4, 512, 512, device="cuda")
    y0 = TS(torch.zeros((16, 512, 512, ops.rupv(4, dt)), device="cuda", dtype=dt), 0, 4)
    y1 = TS(torch.zeros((16, 512, 512, ops.rupv(100, dt)), device="cuda", dtype=dt), 0, 100)
    print(f"nchw_to_nhwc {str(dt)[6:]}: own buffer {timeit(lambda: ops.nchw_to_nhwc(x, y0)):6.1f} us, channels 96..99 of the concat {timeit(lambda: ops.nchw_to_nhwc(x, y1, at=96)):6.1f} us")
    assert torch.equal(y0.buf[..., :4].float(), x.permute(0, 2, 3, 1).to(dt).float()) and torch.equal(y1.buf[..., 96:100].float(), x.permute(0, 2, 3, 1).to(dt).float())
# the final upsample (96 -> 4 x 96 at 16 x 256^2, stored pixel-shuffled into the 100-wide concat buffer) with the network input appended by the
# same launch (unet_conv_desc.ps_tail) against the launch + a second pass
for dt in (torch.float32, torch.bfloat16):
    N, H, Cin, nf = 16, 256, 96, 96
    x = TS(torch.randn((N, H, H, Cin), device="cuda", generator=g).to(dt), 0, Cin)
    w = torch.randn((4 * nf, Cin, 1, 1), device="cuda", generator=g) / Cin ** 0.5
    b = torch.randn(4 * nf, device="cuda", generator=g)
    wp = ops.pack_weights(w, 2, dtype=dt)
    img = torch.rand(N, 4, 2 * H, 2 * H, device="cuda")
    x0 = TS(torch.zeros((N, 2 * H, 2 * H, ops.rupv(4, dt)), device="cuda", dtype=dt), 0, 4)
    ops.nchw_to_nhwc(img, x0)
    Xa = TS(torch.zeros((N, 2 * H, 2 * H, ops.rupv(100, dt)), device="cuda", dtype=dt), 0, nf)
    Xb = TS(torch.zeros((N, 2 * H, 2 * H, ops.rupv(100, dt)), device="cuda", dtype=dt), 0, nf)
    t_plain = timeit(lambda: ops.conv1x1_shuffle(x, wp, Xa, bias=b, relu=True))
    t_put = timeit(lambda: ops.nchw_to_nhwc(img, TS(Xa.buf, 0, Xa.buf.shape[3]), at=96))
    t_tail = timeit(lambda: ops.conv1x1_shuffle(x, wp, Xb, bias=b, relu=True, tail=x0, tail_at=96))
    print(f"upsample {str(dt)[6:]}: launch {t_plain:6.1f} us + second pass {t_put:6.1f} us = {t_plain + t_put:6.1f}; with ps_tail {t_tail:6.1f} us; same buffer: {torch.equal(Xa.buf, Xb.buf)}")
