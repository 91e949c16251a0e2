"""Every distinct conv shape of the cfg2 step (xresnet34 DynamicUnet, 16 tiles of 4x512x512) ALONE on the GPU: forward, input gradient and
weight gradient (kernel + its split reduction), each against its own roofs -- MFMA time at the dtype's dense peak and HBM time of the
algorithmic bytes at 8 TB/s.  Shows which shapes a storage type loses on, without the two streams of the step sharing the chip.
usage: python scripts/layer_bench.py [f32|bf16] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
PEAK = 2500.0 if dt == torch.bfloat16 else 157.3
ES = 2 if dt == torch.bfloat16 else 4
g = torch.Generator(device="cuda").manual_seed(0)
# (Cin, Cout, H_in, ks, stride, launches per step)
LAYERS = [(32, 32, 256, 3, 1, 1), (32, 64, 256, 3, 1, 1), (64, 64, 128, 3, 1, 6), (64, 128, 128, 3, 2, 1), (128, 128, 64, 3, 1, 7),
          (128, 256, 64, 3, 2, 1), (256, 256, 32, 3, 1, 11), (256, 512, 32, 3, 2, 1), (512, 512, 16, 3, 1, 5), (512, 1024, 16, 3, 1, 1),
          (1024, 512, 16, 3, 1, 1), (512, 512, 32, 3, 1, 2), (384, 384, 64, 3, 1, 2), (256, 256, 128, 3, 1, 2), (192, 96, 256, 3, 1, 1),
          (96, 96, 256, 3, 1, 1), (100, 100, 512, 3, 1, 2),
          (512, 1024, 16, 1, 1, 1), (512, 1024, 32, 1, 1, 1), (384, 768, 64, 1, 1, 1), (256, 512, 128, 1, 1, 1), (96, 384, 256, 1, 1, 1)]


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "roof": 0.0}
print(f"{'layer':28s} {'GF':>7s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad':>8s} {'TF':>6s} | {'wgrad':>8s} {'TF':>6s} | {'mfma ms':>7s} {'hbm ms':>7s} | x launches")
for Cin, Cout, H, ks, st, cnt in LAYERS:
    pad = (ks - 1) // 2
    OH = (H + 2 * pad - ks) // st + 1
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, OH, OH, ops.rupv(Cout, dt)), device="cuda", dtype=dt), 0, Cout)
    dy = TS(torch.randn((N, OH, OH, ops.rupv(Cout, dt)), device="cuda", generator=g).to(dt), 0, Cout)
    dx = TS(torch.empty((N, H, H, ops.rupv(Cin, dt)), device="cuda", dtype=dt), 0, Cin)
    w = torch.randn((Cout, Cin, ks, ks), device="cuda", generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    wf, wd = ops.pack_weights(w, 0, dtype=dt), ops.pack_weights(w, 1, dtype=dt)
    ws = torch.empty(ops.wgrad_workspace(x, dy, ks, st, with_bias=True), device="cuda")
    dw = torch.empty_like(w); db = torch.empty(Cout, device="cuda")
    fl = 2.0 * N * OH * OH * Cin * Cout * ks * ks
    t_f = timeit(lambda: ops.conv2d(x, wf, y, ks, st, bias=b, relu=True))
    t_d = timeit(lambda: ops.conv2d_dgrad(dy, wd, dx, ks, st))
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, ks, st, ws, dbias=db))
    by = ES * N * (H * H * Cin + OH * OH * Cout)
    mf, hb = fl / PEAK / 1e9, by / 8e12 * 1e3
    print(f"{Cin:4d}->{Cout:4d} @{H:3d} k{ks} s{st}       {fl / 1e9:7.1f} | {t_f:8.3f} {fl / t_f / 1e9:6.0f} | {t_d:8.3f} {fl / t_d / 1e9:6.0f} | {t_w:8.3f} {fl / t_w / 1e9:6.0f} |"
          f" {mf:7.3f} {hb:7.3f} | x{cnt}", flush=True)
    tot["fwd"] += cnt * t_f; tot["dgrad"] += cnt * t_d; tot["wgrad"] += cnt * t_w; tot["roof"] += cnt * max(mf, hb)
    del x, y, dy, dx, ws
print(f"per step (launch counts applied): fwd {tot['fwd']:.2f} ms, dgrad {tot['dgrad']:.2f} ms, wgrad {tot['wgrad']:.2f} ms; "
      f"sum of max(mfma, hbm) roofs per pass {tot['roof']:.2f} ms")
