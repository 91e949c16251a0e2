"""fp32 3x3 convolutions of the cfg2 decoder with bias + ReLU (isolated launches): python scripts/conv_f32_bias.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import torch
from unet_amd import ops
from unet_amd.ops import TS

N = 16
g = torch.Generator(device="cuda").manual_seed(0)
import unet_amd._lib as L
SW = int(sys.argv[1]) if len(sys.argv) > 1 else -2          # -2: the fp32 256-pixel tile on (default), -1: off
_knobs.set_knob("mfma_shape", SW)
for H, Cin, Cout in [(512, 96, 96), (256, 192, 96), (256, 96, 96), (256, 96, 128), (128, 256, 256), (128, 64, 64), (64, 384, 384), (64, 128, 128), (32, 512, 512), (32, 256, 256), (32, 128, 128), (16, 512, 512), (16, 1024, 512), (16, 512, 1024), (16, 256, 256)]:
    x = TS(torch.randn((N, H, H, ops.rup4(Cin)), device="cuda", generator=g), 0, Cin)
    y = TS(torch.empty((N, H, H, ops.rup4(Cout)), device="cuda"), 0, Cout)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    wp = ops.pack_weights(w, 0)
    ts = []
    for rep in range(3):
        for _ in range(2):
            ops.conv2d(x, wp, y, 3, 1, bias=b, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d(x, wp, y, 3, 1, bias=b, relu=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"[{SW}] {H:4d}^2 {Cin:4d}->{Cout:4d}  {min(ts):7.3f} ms  {fl / min(ts) / 1e9:6.1f} TF  variant {ops.conv2d_variant(x, wp, y, 3, 1)}", flush=True)
