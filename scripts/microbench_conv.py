"""Per-layer micro-benchmark of conv fwd / dgrad / wgrad on the cfg2 layer shapes (B=16), both MFMA shapes."""
import sys, torch
sys.path.insert(0, '.')
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
from unet_amd import ops
from unet_amd._lib import lib
from unet_amd.ops import TS

B = 16
LAYERS = [  # name, H, Cin, Cout, ks
    ("res100", 512, 100, 100, 3), ("u3c1", 256, 192, 96, 3), ("u3c2", 256, 96, 96, 3), ("u2", 128, 256, 256, 3),
    ("u1", 64, 384, 384, 3), ("u0", 32, 512, 512, 3), ("mid", 16, 512, 1024, 3), ("l4", 16, 512, 512, 3), ("l1", 128, 64, 64, 3),
    ("shuf8", 256, 96, 384, 1), ("stem1", 256, 32, 32, 3), ("stem2", 256, 32, 64, 3),
]
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, H, Cin, Cout, ks in LAYERS:
    x = TS(torch.randn(B, H, H, ops.rup4(Cin), device="cuda"), 0, Cin)
    y = TS(torch.randn(B, H, H, ops.rup4(Cout), device="cuda"), 0, Cout)
    w = torch.randn(Cout, Cin, ks, ks, device="cuda")
    wf, wd = ops.pack_weights(w, 0), ops.pack_weights(w, 1)
    dw = torch.empty_like(w)
    ws = torch.empty(ops.wgrad_workspace(x, y, ks, 1), device="cuda")
    gf = 2.0 * B * H * H * Cin * Cout * ks * ks / 1e9
    row = f"{name:8s} {gf:8.1f} GF |"
    for shape in (32, 16):
        _knobs.set_knob("mfma_shape", shape); _knobs.set_knob("wgrad_mfma_shape", 32); _knobs.set_knob("wgrad_narrow", 1 if shape == 16 else 0)
        tf = timeit(lambda: ops.conv2d(x, wf, y, ks, 1))
        td = timeit(lambda: ops.conv2d_dgrad(y, wd, x, ks, 1))
        tw = timeit(lambda: ops.conv2d_wgrad(x, y, dw, ks, 1, ws))
        row += f" mf{shape}: fwd {tf:7.3f} ms {gf/tf:6.1f} TF  dgrad {td:7.3f} ms {gf/td:6.1f} TF  wgrad {tw:7.3f} ms {gf/tw:6.1f} TF |"
    print(row, flush=True)
