import sys, torch
sys.path.insert(0, '.')
from unet_amd import ops
from unet_amd._lib import lib
from unet_amd.ops import TS
B, H, Cin, Cout = 16, 512, 100, 100
def timeit(f, n=4):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for kind in ("randn",):
    x = torch.randn(B, H, H, Cin, device="cuda"); y = torch.randn(B, H, H, Cout, device="cuda")
    if kind == "relu_sparse": x = x.relu(); y = y * (torch.rand_like(y) > 0.5)
    if kind == "tiny": x = x.relu(); y = y * 1e-9 * (torch.rand_like(y) > 0.5)
    xt, yt = TS(x, 0, Cin), TS(y, 0, Cout)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda"); db = torch.empty(Cout, device="cuda")
    ws = torch.empty(200_000_000, device="cuda")
    for narrow in (0, 1, 2, 3, 4, 6, 8):
        lib.unet_set_wgrad_narrow(narrow)
        t0 = timeit(lambda: ops.conv2d_wgrad(xt, yt, dw, 3, 1, ws))
        t1 = timeit(lambda: ops.conv2d_wgrad(xt, yt, dw, 3, 1, ws, dbias=db))
        print(f"{kind:12s} narrow={narrow}: no-bias {t0:.3f} ms  with-bias {t1:.3f} ms", flush=True)
