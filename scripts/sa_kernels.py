"""The fused SelfAttention launches alone on the chip at the cfg2 geometry (B 16, 64 x 64 positions, 48 + 48 + 384 channels):
event-timed per launch, or a plain loop for rocprofv3.  python scripts/sa_kernels.py [iters] [B] [HW]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from unet_amd import ops
from unet_amd.ops import TS

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dp, C = 48, 384
N, CQ = HW * HW, 2 * dp + C
torch.manual_seed(0)
qkv = (torch.randn(B, HW, HW, CQ, device="cuda") * 0.5).to(torch.bfloat16)
dO = torch.randn(B, HW, HW, C, device="cuda").to(torch.bfloat16)
O = torch.zeros(B, HW, HW, C, device="cuda", dtype=torch.bfloat16)
dqkv = torch.zeros_like(qkv)
lse, D = torch.zeros(B * ops.sa_rows(N), device="cuda"), torch.zeros(B * ops.sa_rows(N), device="cuda")
pk, gpk, fpk = (torch.zeros(B * ops.sa_pack_elems(N, c), dtype=torch.bfloat16, device="cuda") for c in (C, dp, dp))
q = TS(qkv, 0, CQ)
L = ops.lib
st = ops._stream


def fwd(): ops.sa_fwd(q, dp, C, pk, TS(O, 0, C), lse)
def bwd(): ops.sa_bwd(q, dp, C, TS(dO, 0, C), pk, gpk, fpk, lse, D, TS(dqkv, 0, CQ))
def packv(): ops.sa_pack(q.sub(2 * dp, C), pk)
def packs():
    ops.sa_pack(q.sub(dp, dp), gpk); ops.sa_pack(q.sub(0, dp), fpk)
def rowdot(): ops.sa_rowdot(TS(dO, 0, C), TS(O, 0, C), D)


packv(); fwd(); rowdot(); packs()
ops.sa_pack(TS(dO, 0, C), pk)
bwd()
torch.cuda.synchronize()
gf_f = 2.0 * N * N * (2 * 64 + C) * B / 1e9          # incl. the maxima pass, query / key lanes padded to 64
gf_b = 2.0 * N * N * (64 * 3 + 2 * C + 64 * 2 + C) * B / 1e9
for name, fn, gf in (("pack H", packv, 0), ("fwd", fwd, gf_f), ("rowdot", rowdot, 0), ("pack G,F", packs, 0), ("bwd (kv + q)", bwd, gf_b)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:14s} {us:9.1f} us" + (f"  {gf / us * 1e3:7.1f} TFLOP/s issued" if gf else ""), flush=True)
