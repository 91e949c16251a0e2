"""fp32 weight gradients of the narrow stem layers: pixel sub-splits of wgrad_kernel (default) against the plain form (unet_tuning.wgrad_narrow = 3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

N = 16
g = torch.Generator(device="cuda").manual_seed(0)
for Cin, Cout, H, st in [(32, 32, 256, 1), (32, 64, 256, 1), (4, 32, 512, 2), (64, 64, 128, 1), (3, 32, 256, 2), (32, 32, 128, 1)]:
    OH = (H + 2 - 3) // st + 1
    x = TS(torch.randn((N, H, H, ops.rup4(Cin)), device="cuda", generator=g), 0, Cin)
    if ops.rup4(Cin) != Cin:
        x.buf[..., Cin:] = 0
    dy = TS(torch.randn((N, OH, OH, ops.rup4(Cout)), device="cuda", generator=g), 0, Cout)
    dw = torch.empty((Cout, Cin, 3, 3), device="cuda"); db = torch.empty(Cout, device="cuda")
    out, ref = [], None
    for narrow in (1, 3, 1, 3):
        with ops.tuning(wgrad_narrow=narrow):
            ws = torch.empty(ops.wgrad_workspace(x, dy, 3, st, with_bias=True), device="cuda")
            for _ in range(3):
                ops.conv2d_wgrad(x, dy, dw, 3, st, ws, dbias=db)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.conv2d_wgrad(x, dy, dw, 3, st, ws, dbias=db)
            b.record(); torch.cuda.synchronize()
        ref = dw.clone() if ref is None else ref
        err = (dw - ref).abs().max().item() / ref.abs().max().item()
        ms = a.elapsed_time(b) / 10
        fl = 2.0 * N * OH * OH * Cin * Cout * 9
        out.append(f"[{narrow}] {ms * 1e3:7.1f} us {fl / ms / 1e9:5.1f} TF ({err:.0e})")
    print(f"{Cin:3d}->{Cout:3d} @{H} s{st}  " + "  ".join(out), flush=True)
