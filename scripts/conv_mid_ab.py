"""bf16 3x3 convolutions of the deep stages (small grids) under several thresholds for the 256-pixel tile: python scripts/conv_mid_ab.py 1 4 6"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import torch
from unet_amd import ops
from unet_amd.ops import TS
import unet_amd._lib as L

settings = [int(v) for v in sys.argv[1:]] or [1, 4, 6]
dt = torch.bfloat16
N = 16
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(16, 512, 512), (16, 1024, 512), (32, 512, 512), (32, 256, 256), (32, 128, 256), (64, 128, 128), (128, 64, 64), (256, 64, 64), (256, 32, 32), (256, 32, 64), (128, 64, 128), (64, 384, 384)]
data = {}
for H, Cin, Cout in shapes:
    x = TS(torch.randn((N, H, H, Cin), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, Cout), device="cuda", dtype=dt), 0, Cout)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    data[(H, Cin, Cout)] = (x, y, ops.pack_weights(w, 0, dtype=dt))


def run(k, n=20):
    x, y, wp = data[k]
    for _ in range(3):
        ops.conv2d(x, wp, y, 3, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.conv2d(x, wp, y, 3, 1)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
for r in range(3):
    for k in shapes:
        for v in settings:
            _knobs.set_knob("bf16_big_tile", v)
            res.setdefault((k, v), []).append(run(k))
            res[(k, v, "var")] = ops.conv2d_variant(data[k][0], data[k][2], data[k][1], 3, 1)
for k in shapes:
    H, Cin, Cout = k
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"{H:3d}^2 {Cin:4d}->{Cout:4d}  " + "   ".join(f"[{v}] {min(res[(k, v)]) * 1e3:6.1f} us {fl / min(res[(k, v)]) / 1e9:5.0f} TF (var {res[(k, v, 'var')]})" for v in settings), flush=True)
_knobs.set_knob("bf16_big_tile", 1)
