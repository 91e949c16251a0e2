"""Isolated bf16 3x3 launches at 16 x 512^2 under several settings of unet_tuning.bf16_big_tile on one box, interleaved (1 = the 256-pixel tile with its
own choice of tiles per workgroup, 100 + n = n tiles per workgroup, 0 = the 128-pixel tile): python scripts/conv_ab.py 1 101 108 [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import torch
from unet_amd import ops
from unet_amd.ops import TS
import unet_amd._lib as L

settings = [int(v) for v in sys.argv[1:] if int(v) >= 0 and (int(v) == 0 or int(v) == 1 or int(v) >= 100)] or [1]
reps = 3
dt = torch.bfloat16
N, H = 16, 512
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(96, 96), (100, 100), (128, 128), (96, 100), (116, 100)]
data = {}
for Cin, Cout in shapes:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, ops.rupv(Cout, dt)), device="cuda", dtype=dt), 0, Cout)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    data[(Cin, Cout)] = (x, y, ops.pack_weights(w, 0, dtype=dt))


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
for r in range(reps):
    for k in shapes:
        for v in settings:
            _knobs.set_knob("bf16_big_tile", v)
            res.setdefault((k, v), []).append(run(k))
for k in shapes:
    fl = 2.0 * N * H * H * k[0] * k[1] * 9
    print(f"{k[0]:4d}->{k[1]:4d} " + "   ".join(f"[{v}] {min(res[(k, v)]):6.3f} ms {fl / min(res[(k, v)]) / 1e9:6.0f} TF" for v in settings), flush=True)
_knobs.set_knob("bf16_big_tile", 1)
