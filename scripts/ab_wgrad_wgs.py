"""bf16 weight gradients of the short-reduction layers: time against the number of workgroups (unet_tuning.wgrad_wgs; 0 = the planner's model)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
N = 16
g = torch.Generator(device="cuda").manual_seed(0)
for Cin, Cout, H, ks in [(64, 64, 128, 3), (128, 128, 64, 3), (256, 256, 32, 3), (512, 512, 16, 3), (512, 512, 32, 3), (384, 384, 64, 3), (256, 256, 128, 3),
                         (32, 32, 256, 3), (32, 64, 256, 3), (100, 100, 512, 3), (192, 96, 256, 3), (512, 1024, 32, 1), (384, 768, 64, 1), (256, 512, 128, 1), (96, 384, 256, 1)]:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    dy = TS(torch.randn((N, H, H, ops.rupv(Cout, dt)), device="cuda", generator=g).to(dt), 0, Cout)
    dw = torch.empty((Cout, Cin, ks, ks), device="cuda"); db = torch.empty(Cout, device="cuda")
    out = []
    ref = None
    for wgs in (0, 128, 192, 256, 384, 512):
        with ops.tuning(wgrad_wgs=wgs):
            ws = torch.empty(ops.wgrad_workspace(x, dy, ks, 1, with_bias=True), device="cuda")
            for _ in range(3):
                ops.conv2d_wgrad(x, dy, dw, ks, 1, ws, dbias=db)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.conv2d_wgrad(x, dy, dw, ks, 1, ws, dbias=db)
            b.record(); torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
        err = (dw - ref).abs().max().item() / ref.abs().max().item()
        out.append(f"{wgs}: {a.elapsed_time(b) / 10 * 1e3:6.1f} us ({err:.0e})")
    print(f"{Cin:4d}->{Cout:4d} @{H:3d} k{ks}  " + "  ".join(out), flush=True)
