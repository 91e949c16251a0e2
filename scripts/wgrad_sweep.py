"""Weight-gradient kernel sweep (isolated launches): python scripts/wgrad_sweep.py [f32|bf16]
The shapes are the three families of cfg2: the general 64 x 64-tiled kernel (256 -> 256, 384 -> 384, 192 -> 192) and the narrow-output
kernel (100 -> 100 at full resolution)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, Cin, Cout) in [(16, 128, 256, 256), (16, 64, 384, 384), (16, 256, 192, 192), (16, 512, 100, 100)]:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    dy = TS(torch.randn((N, H, H, ops.rupv(Cout, dt)), device="cuda", generator=g).to(dt), 0, Cout)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    ws = torch.empty(ops.wgrad_workspace(x, dy, 3, 1), device="cuda")
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, dw, 3, 1, ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.conv2d_wgrad(x, dy, dw, 3, 1, ws)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{N}x{H}x{H} {Cin}->{Cout}: {ms:.3f} ms  {2.0 * N * H * H * Cin * Cout * 9 / ms / 1e9:.0f} TFLOP/s", flush=True)
