"""the 1x1 convs of the cfg2 decoder (PixelShuffle convs, forward and input-gradient forms; isolated launches at batch 16):
UNET_CONV1X1_GEMM=0|1 python scripts/ab_conv1x1.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
N = 16
g = torch.Generator(device="cuda").manual_seed(0)
tag = os.environ.get("UNET_CONV1X1_GEMM", "0")
tot = 0.0
for H, Cin, Cout in [(16, 512, 1024), (32, 512, 1024), (64, 384, 768), (128, 256, 512), (256, 96, 384), (16, 1024, 512), (32, 1024, 512), (64, 768, 384), (128, 512, 256), (256, 384, 96)]:
    x = TS(torch.randn((N, H, H, Cin), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, Cout), device="cuda", dtype=dt), 0, Cout)
    w = torch.randn((Cout, Cin, 1, 1), device="cuda", generator=g) / Cin ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    wp = ops.pack_weights(w, 0, dtype=dt)
    ts = []
    for rep in range(3):
        for _ in range(2):
            ops.conv2d(x, wp, y, 1, 1, bias=b, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d(x, wp, y, 1, 1, bias=b, relu=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * N * H * H * Cin * Cout
    by = N * H * H * (Cin + Cout) * x.buf.element_size()
    tot += min(ts)
    print(f"[gemm1x1 {tag} {sys.argv[1] if len(sys.argv) > 1 else 'f32'}] {H:4d}^2 {Cin:4d}->{Cout:4d}  {min(ts) * 1e3:7.1f} us  {fl / min(ts) / 1e9:6.1f} TF  {by / min(ts) / 1e6:7.1f} GB/s  variant {ops.conv2d_variant(x, wp, y, 1, 1)}", flush=True)
print(f"[gemm1x1 {tag}] sum {tot:.3f} ms")
# the fused final upsample: conv1x1 96 -> 384 + ReLU + PixelShuffle stored into a 100-wide concat slice at 16 x 512^2
x = TS(torch.randn((N, 256, 256, 96), device="cuda", generator=g).to(dt), 0, 96)
X = TS(torch.empty((N, 512, 512, 104 if dt == torch.bfloat16 else 100), device="cuda", dtype=dt), 0, 96)
w = torch.randn((384, 96, 1, 1), device="cuda", generator=g) / 96 ** 0.5
b = torch.randn(384, device="cuda", generator=g)
wp = ops.pack_weights(w, 2, dtype=dt)
ts = []
for rep in range(3):
    for _ in range(2):
        ops.conv1x1_shuffle(x, wp, X, bias=b, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv1x1_shuffle(x, wp, X, bias=b, relu=True)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
print(f"[gemm1x1 {tag} {sys.argv[1] if len(sys.argv) > 1 else 'f32'}] fused upsample 96->384 + shuffle  {min(ts) * 1e3:7.1f} us  checksum {float(X.view().double().sum()):.6e}")
