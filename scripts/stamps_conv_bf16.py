"""Where a workgroup of conv_bf16_t256_kernel spends its life (diagnostic build: UNET_EXTRA_HIPCC_FLAGS=-DUNET_STAMPS python -m unet_amd.build --force).
Per workgroup (wave 0): prologue, main loop, epilogue in shader clocks, cycles inside the per-stage vmcnt waits and the chunk barriers."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from unet_amd import ops
from unet_amd.ops import TS
import unet_amd._lib as L
fn = L.lib.unet_debug_set_stamps
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
N, H = 16, 512
for Cin, Cout in [(96, 96), (100, 100), (128, 128)]:
    x = TS(torch.randn((N, H, H, ops.rupv(Cin, dt)), device="cuda", generator=g).to(dt), 0, Cin)
    y = TS(torch.empty((N, H, H, ops.rupv(Cout, dt)), device="cuda", dtype=dt), 0, Cout)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (Cin * 9) ** 0.5
    wp = ops.pack_weights(w, 0, dtype=dt)
    nblk = (N * H * H // 256 + 7) // 8 * 8
    buf = torch.zeros(nblk * 24, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv2d(x, wp, y, 3, 1)
    assert fn(buf.data_ptr()) == 0
    ops.conv2d(x, wp, y, 3, 1)
    torch.cuda.synchronize()
    fn(None)
    s = buf.cpu().numpy().reshape(-1, 24)
    s = s[s[:, 7] == 1]
    life, pro, loop, epi, waits, bars, nt = s[:, 3] - s[:, 0], s[:, 1] - s[:, 0], s[:, 2], s[:, 18], s[:, 4], s[:, 5], s[:, 19]
    med = lambda a: float(np.median(a))
    span = (s[:, 6].max() - s[:, 6].min()) / 100.0   # realtime: 100 MHz -> us
    print(f"{Cin}->{Cout}: {len(s)} workgroups of {med(nt):.0f} tiles, kernel span {span:.0f} us; per workgroup (median shader clocks): life {med(life):.0f}  first prologue {med(pro):.0f}  "
          f"main loops {med(loop):.0f} (in vmcnt waits {med(waits):.0f}, in chunk barriers {med(bars):.0f})  epilogues (next tile's loads in flight) {med(epi):.0f}", flush=True)
    print("   waits by tap 0..8, folded tail:", " ".join(f"{med(s[:, 8 + i]):.0f}" for i in range(10)), flush=True)
