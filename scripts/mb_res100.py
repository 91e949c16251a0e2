import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
from unet_amd import ops
from unet_amd.ops import TS
B = 16
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, H, Cin, Cout in (("res100", 512, 100, 100), ("u2", 128, 256, 256)):
    x = TS(torch.randn(B, H, H, Cin, device="cuda"), 0, Cin); y = TS(torch.empty(B, H, H, Cout, device="cuda"), 0, Cout)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda"); wf = ops.pack_weights(w, 0)
    gf = 2.0 * B * H * H * Cin * Cout * 9 / 1e9
    t = timeit(lambda: ops.conv2d(x, wf, y, 3, 1))
    print(f"{name}: fwd {t:.3f} ms {gf/t:.1f} TF", flush=True)

# weight gradients of the narrow-output layers (A/B: unet_tuning.wgrad_narrow)
from unet_amd._lib import lib
for name, H, Cin, Cout in (("w100", 512, 100, 100), ("w96", 256, 96, 96), ("w192_96", 256, 192, 96)):
    x = TS(torch.randn(B, H, H, Cin, device="cuda"), 0, Cin); dy = TS(torch.randn(B, H, H, Cout, device="cuda"), 0, Cout)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    gf = 2.0 * B * H * H * Cin * Cout * 9 / 1e9
    for mode in (0, 1):
        _knobs.set_knob("wgrad_narrow", mode)
        ws = torch.empty(ops.wgrad_workspace(x, dy, 3, 1, with_bias=True), device="cuda")
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, ws))
        print(f"{name} narrow={mode}: wgrad {t:.3f} ms {gf/t:.1f} TF", flush=True)
