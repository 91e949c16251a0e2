"""fp32 1x1 weight gradients with few input (or output) channels: the 128 x 128-blocked GEMM kernel (unet_tuning.wgrad_1x1 = 1) against the
64 x 64-blocked general kernel (0).  Shapes: SelfAttention's dF (48 x 4096 per image), the encoder's identity-path convs, PixelShuffle convs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS
g = torch.Generator(device="cuda").manual_seed(0)
for Cin, Cout, H, N in [(48, 4096, 64, 1), (4096, 48, 64, 1), (64, 128, 64, 16), (128, 256, 32, 16), (256, 512, 16, 16), (96, 384, 256, 16), (384, 4096, 64, 1), (480, 384, 64, 16), (384, 480, 64, 16)]:
    x = TS(torch.randn((N, H, H, Cin), device="cuda", generator=g), 0, Cin)
    dy = TS(torch.randn((N, H, H, Cout), device="cuda", generator=g), 0, Cout)
    dw = torch.empty((Cout, Cin, 1, 1), device="cuda")
    out, ref = [], None
    for mode in (1, 0, 1, 0):
        with ops.tuning(wgrad_1x1=mode):
            ws = torch.empty(ops.wgrad_workspace(x, dy, 1, 1), device="cuda")
            for _ in range(3): ops.conv2d_wgrad(x, dy, dw, 1, 1, ws)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): ops.conv2d_wgrad(x, dy, dw, 1, 1, ws)
            b.record(); torch.cuda.synchronize()
        ref = dw.clone() if ref is None else ref
        ms = a.elapsed_time(b) / 10
        out.append(f"[{mode}] {ms * 1e3:7.1f} us {2.0 * N * H * H * Cin * Cout / ms / 1e9:5.1f} TF ({(dw - ref).abs().max().item() / ref.abs().max().item():.0e})")
    print(f"{Cin:4d}->{Cout:4d} @{H} x{N}  " + "  ".join(out), flush=True)
