"""Relative L2 error of the conv forward / input-gradient / weight-gradient kernels against fp64, next to torch-CPU fp32 (oneDNN), at the
shapes of the network: the sqrt(K) growth of the k-ordered fp32 MFMA chain quoted in DESIGN.md section 4 and tests/test_configs_gpu.py.
Run on the GPU box: python scripts/accuracy_vs_fp64.py"""
import sys, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from unet_amd import ops
from tests.util import to_ts, empty_ts, from_ts
torch.set_num_threads(16)
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()
g = torch.Generator().manual_seed(0)
for (N, H, W, Cin, Cout, ks) in ((2, 8, 8, 512, 1024, 3), (2, 8, 8, 1024, 512, 3), (2, 8, 8, 512, 1024, 1), (2, 16, 16, 512, 512, 3), (2, 32, 32, 256, 256, 3),
                                 (2, 256, 256, 99, 99, 3), (2, 128, 128, 192, 96, 3)):
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g)
    pad = (ks - 1) // 2
    y64 = F.conv2d(x.double(), w.double(), None, padding=pad)
    y32 = F.conv2d(x, w, None, padding=pad)
    xt, yt = to_ts(x), empty_ts(N, H, W, Cout)
    ops.conv2d(xt, ops.pack_weights(w.cuda(), 0), yt, ks, 1)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=pad)
    dx32 = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=pad)
    dyt, dxt = to_ts(dy), empty_ts(N, H, W, Cin)
    ops.conv2d_dgrad(dyt, ops.pack_weights(w.cuda(), 1), dxt, ks, 1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=pad)
    dw32 = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=pad)
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    n = ops.wgrad_workspace(xt, dyt, ks, 1)
    ops.conv2d_wgrad(xt, dyt, dw, ks, 1, torch.empty(n, device="cuda"))
    torch.cuda.synchronize()
    print(f"{(N,H,W,Cin,Cout,ks)}: fwd hip {rel(from_ts(yt), y64):.2e} cpu {rel(y32, y64):.2e} | dgrad hip {rel(from_ts(dxt), dx64):.2e} cpu {rel(dx32, dx64):.2e} | wgrad hip {rel(dw.cpu(), dw64):.2e} cpu {rel(dw32, dw64):.2e}")
