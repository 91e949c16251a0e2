"""conv forward at 16x512x512, Cin=100, varying Cout: how much of the 4-vs-3 tile imbalance at Cout=100 is already hidden"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS
B, H, Cin = 16, 512, 100
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = TS(torch.randn(B, H, H, Cin, device="cuda"), 0, Cin)
for Cout in (80, 96, 100, 112, 128):
    y = TS(torch.empty(B, H, H, Cout, device="cuda"), 0, Cout)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda"); wf = ops.pack_weights(w, 0)
    t = timeit(lambda: ops.conv2d(x, wf, y, 3, 1))
    tiles = -(-Cout // 16)
    print(f"Cout {Cout:4d} ({tiles} tiles): {t:.3f} ms  {t / tiles:.3f} ms/tile  {2.0 * B * H * H * Cin * Cout * 9 / 1e9 / t:.1f} TF", flush=True)
