"""A/B of the fp32 4-channel sliver in conv_bf16_t256_kernel<7, 32, float, true> (isolated 16 x 512^2 launches of the final ResBlock pair):
UNET_T256_SLIVER=0|1 python scripts/ab_sliver.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

N, H = 16, 512
g = torch.Generator(device="cuda").manual_seed(0)
tag = os.environ.get("UNET_T256_SLIVER", "1")
for Cin, Cout, res in [(100, 100, False), (100, 100, True), (96, 96, False), (100, 96, False), (96, 100, False)]:
    x = TS(torch.randn((N, H, H, ops.rup4(Cin)), device="cuda", generator=g), 0, Cin)
    y = TS(torch.empty((N, H, H, ops.rup4(Cout)), device="cuda"), 0, Cout)
    r = TS(torch.randn((N, H, H, ops.rup4(Cout)), device="cuda", generator=g), 0, Cout) if res else None
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    wp = ops.pack_weights(w, 0)
    ts = []
    for rep in range(3):
        for _ in range(2):
            ops.conv2d(x, wp, y, 3, 1, bias=b, relu=True, res=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.conv2d(x, wp, y, 3, 1, bias=b, relu=True, res=r)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"[sliver {tag}] {Cin:4d}->{Cout:4d} res={int(res)}  {min(ts):7.3f} ms  {fl / min(ts) / 1e9:6.1f} TF  checksum {float(y.view().double().sum()):.6e}", flush=True)
