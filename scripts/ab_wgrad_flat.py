"""A/B of the 4-row sliver in wgrad_flat_kernel<6, 5, true> against the seven-tile form <7, 5> (unet_tuning.wgrad_narrow = 2), isolated
16 x 512^2 launches of the final ResBlock pair's weight gradients (kernel + split reduction + bias reduction as the step issues them)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

N, H = 16, 512
g = torch.Generator(device="cuda").manual_seed(0)
for Cin, Cout in [(100, 100), (96, 100), (100, 96)]:
    x = TS(torch.randn((N, H, H, ops.rup4(Cin)), device="cuda", generator=g), 0, Cin)
    dy = TS(torch.randn((N, H, H, ops.rup4(Cout)), device="cuda", generator=g), 0, Cout)
    res = {}
    for narrow in (1, 2, 1, 2):
        with ops.tuning(wgrad_narrow=narrow):
            ws = torch.empty(ops.wgrad_workspace(x, dy, 3, 1), device="cuda")
            dw = torch.empty((Cout, Cin, 3, 3), device="cuda"); db = torch.empty(Cout, device="cuda")
            for _ in range(2):
                ops.conv2d_wgrad(x, dy, dw, 3, 1, ws, dbias=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.conv2d_wgrad(x, dy, dw, 3, 1, ws, dbias=db)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        fl = 2.0 * N * H * H * Cin * Cout * 9
        res.setdefault(narrow, []).append((ms, dw.double().sum().item(), dw.clone()))
        print(f"[wgrad_narrow {narrow}] {Cin:4d}->{Cout:4d}  {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF  checksum {dw.double().sum().item():.9e}", flush=True)
    a, b = res[1][0][2], res[2][0][2]
    print(f"   max |sliver - seven tiles| / max |dw| = {(a - b).abs().max().item() / b.abs().max().item():.2e}")
