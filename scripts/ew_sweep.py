"""HBM-bound kernels of the cfg2 step in isolation, fp32 and bf16 storage: algorithmic bytes (every tensor once) / launch time.
python scripts/ew_sweep.py [bf16|f32 ...]"""
import sys
import torch
sys.path.insert(0, '.')
from unet_amd import ops
from unet_amd.ops import TS

B = 16


def t(N, H, W, C, dt):
    return ops.new_act(N, H, W, C, "cuda", zero=True, dtype=dt)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(name, ms, nbytes):
    print(f"{name:52s} {ms * 1e3:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / ms / 1e6:8.0f} GB/s", flush=True)


for dts in (sys.argv[1:] or ["bf16", "f32"]):
    dt = torch.bfloat16 if dts == "bf16" else torch.float32
    es = 2 if dts == "bf16" else 4
    print(f"---- {dts}")
    # decoder shuffles: (h, Cu, blur)
    for h, Cu, blur in ((256, 96, False), (128, 128, True), (64, 192, True), (32, 256, True), (16, 256, True)):
        yc, X = t(B, h, h, 4 * Cu, dt), t(B, 2 * h, 2 * h, Cu, dt)
        dyc = t(B, h, h, 4 * Cu, dt)
        n = B * h * h * 4 * Cu * es
        report(f"shuffle_blur     h={h} Cu={Cu} blur={blur}", timeit(lambda: ops.shuffle_blur(yc, X, blur)), 2 * n)
        report(f"shuffle_blur_bwd h={h} Cu={Cu} blur={blur}", timeit(lambda: ops.shuffle_blur_bwd(X, yc, dyc, blur)), 3 * n)
    # encoder BatchNorm family: (H, C)
    for H, C in ((256, 32), (256, 64), (128, 64), (64, 128), (32, 256), (16, 512)):
        x, y, d, dx = t(B, H, H, C, dt), t(B, H, H, C, dt), t(B, H, H, C, dt), t(B, H, H, C, dt)
        v = lambda: torch.ones(C, device="cuda")
        sc, sh, mean, inv, g, c1, c2 = v(), v(), v(), v(), v(), v(), v()
        P = x.P
        part = torch.zeros(2 * ops.bn_stats_rows(P) * C + 1024, device="cuda")
        n = P * C * es
        report(f"bn_stats         H={H} C={C}", timeit(lambda: ops.bn_stats(x, part)), n)
        report(f"affine_act relu  H={H} C={C}", timeit(lambda: ops.affine_act(x, y, sc, sh, relu=True)), 2 * n)
        report(f"affine_act +res  H={H} C={C}", timeit(lambda: ops.affine_act(x, y, sc, sh, x2=d, relu=True)), 3 * n)
        report(f"bn_bwd_reduce    H={H} C={C}", timeit(lambda: ops.bn_bwd_reduce(d, y, x, mean, inv, part)), 3 * n)
        report(f"bn_bwd_apply     H={H} C={C}", timeit(lambda: ops.bn_bwd_apply(d, y, x, mean, inv, g, c1, c2, dx)), 4 * n)
    x, y = t(B, 256, 256, 64, dt), t(B, 128, 128, 64, dt)
    idx = torch.zeros(B * 128 * 128 * 64, dtype=torch.uint8, device="cuda")
    dxm = t(B, 256, 256, 64, dt)
    n = B * 128 * 128 * 64
    report("maxpool 256->128 C=64", timeit(lambda: ops.maxpool(x, y, idx)), 4 * n * es + n * es + n)
    report("maxpool_bwd", timeit(lambda: ops.maxpool_bwd(y, idx, dxm)), n * es + n + 4 * n * es)
    xin = torch.rand(B, 4, 512, 512, device="cuda")
    x0 = t(B, 512, 512, 4, dt)
    report("nchw_to_nhwc 4ch 512", timeit(lambda: ops.nchw_to_nhwc(xin, x0)), B * 4 * 512 * 512 * (4 + es))
    a, b = t(B, 128, 128, 64, dt), t(B, 128, 128, 64, dt)
    report("copy_slice accumulate 128^2 C=64", timeit(lambda: ops.copy_slice(a, b, accumulate=True)), 3 * B * 128 * 128 * 64 * es)
