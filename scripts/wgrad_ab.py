"""bf16 3x3 weight gradients at 16 x 512^2 / 256^2 / 128^2: wgrad_bf16_k4_kernel (on) against wgrad_bf16_kernel<32,1,3> (off), interleaved on one box"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import torch
from unet_amd import ops
from unet_amd.ops import TS
import unet_amd._lib as L

dt = torch.bfloat16
N = 16
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(512, 100, 100, 3), (512, 96, 96, 3), (256, 192, 96, 3), (128, 256, 256, 3), (64, 384, 384, 3), (256, 96, 384, 1), (128, 256, 512, 1), (64, 384, 768, 1), (512, 100, 5, 1)]
data = {}
for H, Cin, Cout, ks in shapes:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    dy = TS(torch.randn((N, H, H, ops.rupv(Cout, dt)), device="cuda", generator=g).to(dt), 0, Cout)
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    ws = torch.empty(ops.wgrad_workspace(x, dy, ks, 1, with_bias=True), device="cuda")
    data[(H, Cin, Cout, ks)] = (x, dy, dw, ws, torch.empty(Cout, device="cuda"))


def run(k, n=10):
    x, dy, dw, ws, db = data[k]
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, dw, k[3], 1, ws, dbias=db)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.conv2d_wgrad(x, dy, dw, k[3], 1, ws, dbias=db)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
ref = {}
for r in range(3):
    for k in shapes:
        for v in (-2, -1):
            _knobs.set_knob("wgrad_mfma_shape", v)
            res.setdefault((k, v), []).append(run(k))
            if r == 0:
                ref[(k, v)] = data[k][2].clone()
for k in shapes:
    H, Cin, Cout, ks = k
    fl = 2.0 * N * H * H * Cin * Cout * ks * ks
    d = (ref[(k, -2)] - ref[(k, -1)]).abs().max().item() / ref[(k, -1)].abs().max().item()
    print(f"{H:4d}^2 {Cin:4d}->{Cout:4d} k{ks}  k4 {min(res[(k, -2)]):6.3f} ms {fl / min(res[(k, -2)]) / 1e9:6.0f} TF   2x2 {min(res[(k, -1)]):6.3f} ms {fl / min(res[(k, -1)]) / 1e9:6.0f} TF   (incl. bias gradient and reduce)  rel diff {d:.1e}", flush=True)
_knobs.set_knob("wgrad_mfma_shape", -2)
