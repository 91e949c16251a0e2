"""What the bias gradient costs inside the weight-gradient launches (it is computed by the workgroups of the first input-channel block):
python scripts/wgrad_bias_cost.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
N = 16
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(512, 100, 100, 3), (512, 96, 96, 3), (256, 192, 96, 3), (128, 256, 256, 3), (64, 384, 384, 3), (256, 96, 384, 1), (128, 256, 512, 1), (512, 100, 5, 1)]
for H, Cin, Cout, ks in shapes:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    dy = TS(torch.randn((N, H, H, ops.rupv(Cout, dt)), device="cuda", generator=g).to(dt), 0, Cout)
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    db = torch.empty(Cout, device="cuda")
    ws = torch.empty(ops.wgrad_workspace(x, dy, ks, 1, with_bias=True), device="cuda")
    out = []
    for bias in (None, db):
        ts = []
        for rep in range(3):
            for _ in range(2):
                ops.conv2d_wgrad(x, dy, dw, ks, 1, ws, dbias=bias)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.conv2d_wgrad(x, dy, dw, ks, 1, ws, dbias=bias)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 10)
        out.append(min(ts))
    print(f"{H:4d}^2 {Cin:4d}->{Cout:4d} k{ks}   without bias {out[0]:6.3f} ms   with {out[1]:6.3f} ms   (+{(out[1] / out[0] - 1) * 100:4.1f} %)", flush=True)
