"""The stem's first conv (Cin -> 32, 3x3 stride 2) alone on the GPU: conv3x3_smallcin_kernel against the implicit-GEMM kernels
(unet_tuning.conv_smallcin = 0), both storage types, batch 16 and 1 of 512^2 tiles.  usage: python scripts/ab_conv_smallcin.py"""
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
    for N, Cin, Cout, H, st in [(16, 4, 32, 512, 2), (1, 4, 32, 512, 2), (16, 3, 32, 512, 2), (16, 8, 32, 512, 2), (16, 4, 32, 256, 2), (16, 8, 64, 256, 1)]:
        OH = (H + 2 - 3) // st + 1
        x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
        w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
        b = torch.randn(Cout, device="cuda", generator=g)
        wf = ops.pack_weights(w, 0, dtype=dt)
        res = []
        for on in (1, 0):
            with ops.tuning(conv_smallcin=on):
                y = TS(torch.empty((N, OH, OH, ops.rupv(Cout, dt)), device="cuda", dtype=dt), 0, Cout)
                var = ops.conv2d_variant(x, wf, y, 3, st)
                t = timeit(lambda: ops.conv2d(x, wf, y, 3, st, bias=b, relu=True))
                res.append((var, t, y.buf.float()))
        by = es * N * (H * H * ops.rupv(Cin, dt) + OH * OH * Cout)
        d = (res[0][2] - res[1][2]).abs().max().item()
        print(f"{str(dt)[6:]:9s} N{N:2d} {Cin}->{Cout} @{H} s{st}: direct {res[0][1]:7.1f} us ({by / res[0][1] / 1e6:5.2f} TB/s, variant {res[0][0]})  "
              f"igemm {res[1][1]:7.1f} us (variant {res[1][0]})  max |diff| {d:.2e}", flush=True)
