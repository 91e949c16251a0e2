"""Conv-kernel sweep on one geometry: where do the 100-channel layers of cfg2 lose against the 96 / 128-channel ones?
usage: python scripts/conv_sweep.py [f32|bf16] [H] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = torch.Generator(device="cuda").manual_seed(0)
for Cin, Cout in [(96, 96), (100, 96), (96, 100), (100, 100), (112, 96), (112, 112), (128, 128), (116, 100), (104, 104)]:
    ci, co = ops.rupv(Cin, dt), ops.rupv(Cout, dt)
    x = TS(torch.randn((N, H, H, ci), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, co), device="cuda", dtype=dt), 0, Cout)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    wp = ops.pack_weights(w, 0, dtype=dt)
    for _ in range(3):
        ops.conv2d(x, wp, y, 3, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv2d(x, wp, y, 3, 1)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f"{Cin:4d}->{Cout:4d}  variant {ops.conv2d_variant(x, wp, y, 3, 1)}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF", flush=True)
