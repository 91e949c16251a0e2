"""HBM write amplification of isolated bf16 3x3 launches at 16 x 512^2 (run under rocprofv3 --pmc WRITE_SIZE --kernel-trace, dispatches in
the printed order): plain, with bias + residual + ReLU, and the same into a 128-channel pixel stride."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16
N, H = 16, 512
g = torch.Generator(device="cuda").manual_seed(0)
cases = [(100, 100, None, False), (100, 100, None, True), (100, 100, 128, False), (100, 100, 128, True), (96, 96, None, True), (128, 128, None, True)]
for Cin, Cout, cs, epi in cases:
    cs_in = ops.rupv(Cin, dt)
    cs_out = cs or ops.rupv(Cout, dt)
    x = TS(torch.randn((N, H, H, cs_in), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, cs_out), device="cuda", dtype=dt), 0, Cout)
    r = TS(torch.randn((N, H, H, cs_out), device="cuda", generator=g).to(dt), 0, Cout) if epi else None
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    wp = ops.pack_weights(w, 0, dtype=dt)
    b = torch.randn(Cout, device="cuda", generator=g) if epi else None
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv2d(x, wp, y, 3, 1, bias=b, res=r, relu=epi)
    a.record()
    ops.conv2d(x, wp, y, 3, 1, bias=b, res=r, relu=epi)
    e.record(); torch.cuda.synchronize()
    print(f"{Cin}->{Cout} pixel stride {cs_out} epilogue {epi}: {a.elapsed_time(e):.3f} ms, output {N * H * H * Cout * 2 / 1e6:.0f} MB", flush=True)
    del x, y, r
