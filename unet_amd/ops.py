"""Thin Python wrappers over the C ABI: torch tensors in, HIP launches out.

PyTorch is used for device memory and streams only.  Activations are fp32 NHWC
buffers; ``TS`` names a channel slice ``buf[..., co:co+C]`` of a buffer with
physical channel stride ``cs`` (this is how ``torch.cat`` disappears).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from ._lib import ConvDesc, WgradDesc, check, lib


# ---- kernel-selection switches (unet_tuning).  The LIBRARY keeps no state: every descriptor carries a pointer to the struct that is current
# on this side when it is built -- None (the defaults) unless a `with ops.tuning(...)` block is active in this thread.
import contextlib
import threading

_TUNE = threading.local()


@contextlib.contextmanager
def tuning(**fields):
    """`with ops.tuning(mfma_shape=32, conv_splitk=0): ...` -- launches issued by this thread inside the block carry these switches
    (A/B measurements; the tests cross-check independently written kernel families this way).  Blocks nest: inner fields override outer."""
    prev = getattr(_TUNE, "cur", None)
    cur = L.Tuning.default() if prev is None else L.Tuning.from_buffer_copy(prev)
    for k, v in fields.items():
        if k not in dict(L.Tuning._fields_):
            raise KeyError(f"unet_tuning has no field {k!r}")
        setattr(cur, k, int(v))
    _TUNE.cur = cur
    try:
        yield cur
    finally:
        _TUNE.cur = prev


def set_tuning(**fields) -> None:
    """the switches of THIS thread's launches from now on (outside `with tuning(...)` blocks); no arguments: back to the defaults.
    Python-side convenience for scripts and try / finally style tests -- the library itself stays stateless."""
    if not fields:
        _TUNE.cur = None
        return
    cur = getattr(_TUNE, "cur", None)
    cur = L.Tuning.default() if cur is None else L.Tuning.from_buffer_copy(cur)
    for k, v in fields.items():
        if k not in dict(L.Tuning._fields_):
            raise KeyError(f"unet_tuning has no field {k!r}")
        setattr(cur, k, int(v))
    _TUNE.cur = cur


def set_knob(name: str, v: int) -> int:
    """the value conventions of the setters the ABI had up to version 5 (unet_set_<name>), mapped onto unet_tuning fields"""
    v = int(v)
    if name == "mfma_shape":
        set_tuning(**({"f32_big_tile": int(v == -2)} if v < 0 else {"mfma_shape": v}))
    elif name == "wgrad_mfma_shape":
        set_tuning(**({"wgrad_bf16_k4": int(v == -2)} if v < 0 else {"wgrad_mfma_shape": v}))
    elif name == "bf16_big_tile":
        set_tuning(t256_tiles_per_wg=v - 100 if v >= 100 else 0, bf16_big_tile=(v if 2 <= v < 100 else int(bool(v))))
    elif name in ("conv_splitk", "wgrad_narrow", "wgrad_1x1", "t256_sliver", "conv1x1_gemm"):
        set_tuning(**{name: max(v, 0)})
    else:
        raise KeyError(name)
    return 0


def _tuning_ptr():
    cur = getattr(_TUNE, "cur", None)
    return None if cur is None else C.pointer(cur)


def wgrad_tuning_key():
    """the values of the switches a weight-gradient plan depends on (None = the defaults): cache key of planned workspace sizes"""
    cur = getattr(_TUNE, "cur", None)
    return None if cur is None else (cur.wgrad_mfma_shape, cur.wgrad_bf16_k4, cur.wgrad_1x1, cur.wgrad_narrow, cur.plan_batch, cur.wgrad_wgs)


# The launch stream of this thread as a raw hipStream_t.  torch.cuda.current_stream() builds a Stream object through several Python layers
# (device-index parsing, is_available(), an os.environ lookup): 8 us of a ~16 us host budget per launch -- scripts/host_profile.py showed the
# cfg1 step HOST-bound (issue time 6.95 ms = wall time), 29 % of it in that call.  The C entry points below are what it ends in.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


# a raw stream that replaces THIS THREAD's current one for the launches it issues while it is set (the side stream of the weight gradients:
# entering torch.cuda.stream(...) costs four current_stream() round trips per weight gradient).  Thread-local like the tuning state: a launch
# from another thread (a second model, a loader) never lands on somebody else's side stream.
class _StreamOverride(threading.local):
    raw: Optional[int] = None


_OVERRIDE = _StreamOverride()


def set_stream_override(raw: Optional[int]) -> None:
    _OVERRIDE.raw = raw


def _stream() -> int:
    o = _OVERRIDE.raw
    if o is not None:
        return o
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _device_index() -> int:
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def rup4(c: int) -> int:
    return (c + 3) // 4 * 4


def vec_of(dtype) -> int:
    """channels per 16-byte access: channel strides / offsets of a tensor are multiples of this"""
    return 8 if dtype == torch.bfloat16 else 4


def rupv(c: int, dtype) -> int:
    v = vec_of(dtype)
    return (c + v - 1) // v * v


@dataclass
class TS:
    """Channel slice of an NHWC buffer: fp32 (the parity path) or bf16 storage (BASELINE configs[1] variant)."""
    buf: torch.Tensor   # [N, H, W, cs] contiguous fp32 | bf16
    co: int
    C: int

    def __post_init__(self):
        assert self.buf.dtype in (torch.float32, torch.bfloat16) and self.buf.is_contiguous() and self.buf.dim() == 4
        v = vec_of(self.buf.dtype)
        assert self.co % v == 0 and self.cs % v == 0 and self.co + self.C <= self.cs, (self.co, self.C, self.cs)

    @property
    def bf16(self) -> bool: return self.buf.dtype == torch.bfloat16

    @property
    def cs(self) -> int: return self.buf.shape[3]
    @property
    def N(self) -> int: return self.buf.shape[0]
    @property
    def H(self) -> int: return self.buf.shape[1]
    @property
    def W(self) -> int: return self.buf.shape[2]
    @property
    def P(self) -> int: return self.buf.shape[0] * self.buf.shape[1] * self.buf.shape[2]
    @property
    def ptr(self) -> int: return self.buf.data_ptr()

    def view(self) -> torch.Tensor:
        return self.buf[..., self.co:self.co + self.C]

    def sub(self, off: int, C: int) -> "TS":
        return TS(self.buf, self.co + off, C)


def new_act(N, H, W, C, device, zero=False, dtype=torch.float32) -> TS:
    cs = rupv(C, dtype)
    need_zero = zero or cs != C
    buf = (torch.zeros if need_zero else torch.empty)((N, H, W, cs), dtype=dtype, device=device)
    return TS(buf, 0, C)


def _fn(name: str, t: "TS"):
    """the entry point for t's storage type: unet_<name> (fp32) or unet_<name>_bf16"""
    return getattr(lib, f"unet_{name}_bf16" if t.bf16 else f"unet_{name}")


def _need_f32(what: str, *ts):
    for t in ts:
        if t is not None and t.bf16:
            raise L.UnetHipError(f"{what}: not available with bf16 storage (fp32 parity path only)")


def full(buf: torch.Tensor, C: Optional[int] = None) -> TS:
    return TS(buf, 0, buf.shape[3] if C is None else C)


# ------------------------------------------------------------------ conv

def conv_out_hw(H, W, ks, stride):
    pad = (ks - 1) // 2
    return (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1


def pack_weights(w: torch.Tensor, mode: int, out: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    """w: [Cout, Cin, ks, ks] contiguous fp32 master parameter.  mode 0 = forward image, 1 = dgrad image; dtype = storage type of
    the packed image (bf16: rounded once here, products accumulate in fp32)."""
    Cout, Cin, ks, _ = w.shape
    assert w.is_contiguous() and w.dtype == torch.float32
    if dtype == torch.bfloat16:
        n = lib.unet_pack_weights_size_bf16(Cout, Cin, ks, mode)
        if out is None or out.dtype != dtype:
            out = torch.empty(n, dtype=dtype, device=w.device)
        assert out.numel() >= n
        check(lib.unet_pack_weights_bf16(w.data_ptr(), out.data_ptr(), Cout, Cin, ks, mode, _stream()), "pack_weights_bf16")
        return out
    n = lib.unet_pack_weights_size(Cout, Cin, ks, mode)
    if out is None or out.dtype != dtype:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    assert out.numel() >= n
    check(lib.unet_pack_weights(w.data_ptr(), out.data_ptr(), Cout, Cin, ks, mode, _stream()), "pack_weights")
    return out


def pack_jobs(jobs, bf16: bool, device, cache: Optional[dict] = None):
    """Packed filter images of several convs in ONE launch (unet_pack_batch_run).  jobs: [(w [Cout,Cin,ks,ks] fp32 master parameter, packed
    image tensor, mode 0 | 1, out_scale [Cout] fp32 or None)].  The device job table holds raw addresses: `cache` (a dict owned by the
    caller) keeps tables keyed by every address in them."""
    if not jobs:
        return
    key = tuple((w.data_ptr(), wp.data_ptr(), mode, 0 if sc is None else sc.data_ptr()) for w, wp, mode, sc in jobs) + (bf16,)
    ent = None if cache is None else cache.get(key)
    if ent is None:
        arr = (L.PackJob * len(jobs))()
        for j, (w, wp, mode, sc) in zip(arr, jobs):
            assert w.is_contiguous() and w.dtype == torch.float32 and (sc is None or (sc.dtype == torch.float32 and sc.numel() >= w.shape[0]))
            j.w, j.wp = w.data_ptr(), wp.data_ptr()
            j.Cout, j.Cin, j.ks, j.mode = w.shape[0], w.shape[1], w.shape[2], mode
            j.out_scale = None if sc is None else sc.data_ptr()
        dt = L.BF16 if bf16 else L.F32
        host = torch.zeros(int(lib.unet_pack_batch_table_bytes(len(jobs))), dtype=torch.uint8)
        blocks = C.c_uint(0)
        check(lib.unet_pack_batch_build(arr, len(jobs), dt, host.data_ptr(), C.byref(blocks)), "pack_batch_build")
        ent = (host.to(device), len(jobs), int(blocks.value), dt)
        if cache is not None:
            cache[key] = ent
    table, n, blocks, dt = ent
    check(lib.unet_pack_batch_run(table.data_ptr(), n, blocks, dt, _stream()), "pack_batch_run")


def pack_weights_strided(w_base_ptr: int, so: int, sr: int, O: int, R: int, out: torch.Tensor) -> torch.Tensor:
    """packed 1x1 filter image whose (out o, reduction r) element is the element at w_base_ptr + es * (o*so + r*sr) of an activation buffer
    of out's storage type (es = 4 fp32 / 2 bf16)"""
    if out.dtype == torch.bfloat16:
        assert out.numel() >= lib.unet_pack_weights_size_bf16(O, R, 1, 0)
        check(lib.unet_pack_weights_strided_bf16(w_base_ptr, so, sr, out.data_ptr(), O, R, _stream()), "pack_weights_strided_bf16")
        return out
    assert out.numel() >= lib.unet_pack_weights_size(O, R, 1, 0)
    check(lib.unet_pack_weights_strided(w_base_ptr, so, sr, out.data_ptr(), O, R, _stream()), "pack_weights_strided")
    return out


def pack_size(O: int, R: int, dtype) -> int:
    """elements of the packed 1x1 image of an [O, R] operand in storage type dtype"""
    return int((lib.unet_pack_weights_size_bf16 if dtype == torch.bfloat16 else lib.unet_pack_weights_size)(O, R, 1, 0))


def row_softmax(x: "TS", y: "TS"):
    """y = softmax over the C channels of every pixel of x; x (the logits) is fp32, y fp32 or bf16"""
    _need_f32("row_softmax (logits)", x)
    fn = lib.unet_row_softmax_bf16 if y.bf16 else lib.unet_row_softmax
    check(fn(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.P, x.C, _stream()), "row_softmax")


def row_softmax_bwd(y: "TS", dy: "TS", dx: "TS"):
    """dx = y * (dy - sum_c y dy); y and dx share a storage type (fp32 | bf16), dy is fp32"""
    _need_f32("row_softmax_bwd (dy)", dy)
    assert y.bf16 == dx.bf16
    fn = lib.unet_row_softmax_bwd_bf16 if y.bf16 else lib.unet_row_softmax_bwd
    check(fn(y.ptr, y.cs, y.co, dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, y.P, y.C, _stream()), "row_softmax_bwd")


# ---- fused SelfAttention for bf16 storage (csrc/attention.hip): the N x N matrix stays in registers
def sa_fused_supported(dp: int, C: int) -> bool:
    return bool(lib.unet_sa_fused_supported(int(dp), int(C)))


def sa_pack_elems(N: int, cc: int) -> int:
    return int(lib.unet_sa_pack_elems(int(N), int(cc)))


def sa_pack(x: "TS", out: torch.Tensor):
    """positions-innermost image of the channel slice x (rows = the H * W positions of each image): the operand of a product that sums over
    positions"""
    assert x.bf16 and out.dtype == torch.bfloat16 and out.numel() >= x.N * sa_pack_elems(x.H * x.W, x.C)
    check(lib.unet_sa_pack_bf16(x.ptr, x.cs, x.co, x.C, x.N, x.H * x.W, out.data_ptr(), _stream()), "sa_pack")


def sa_rows(N: int) -> int:
    """row count per image of the lse / D vectors of the fused attention kernels (whole 64-position blocks)"""
    return (N + 63) // 64 * 64


def sa_fwd(qkv: "TS", dp: int, C: int, vpack: torch.Tensor, O: "TS", lse: torch.Tensor):
    """O_j = sum_i softmax_i(G_j . F_i) H_i and lse_j for every image of the fused QKV buffer (F at channel 0, G at dp, H at 2 dp)"""
    assert qkv.bf16 and O.bf16 and qkv.co == 0 and lse.dtype == torch.float32 and O.C == C and O.P == qkv.P
    assert lse.numel() >= qkv.N * sa_rows(qkv.H * qkv.W) and vpack.numel() >= qkv.N * sa_pack_elems(qkv.H * qkv.W, C)
    check(lib.unet_sa_fwd_bf16(qkv.ptr, qkv.cs, dp, C, qkv.N, qkv.H * qkv.W, vpack.data_ptr(), O.ptr, O.cs, O.co, lse.data_ptr(), _stream()), "sa_fwd")


def sa_rowdot(a: "TS", o: "TS", D: torch.Tensor):
    assert a.bf16 and o.bf16 and a.C == o.C and a.P == o.P and D.dtype == torch.float32 and D.numel() >= a.N * sa_rows(a.H * a.W)
    check(lib.unet_sa_rowdot_bf16(a.ptr, a.cs, a.co, o.ptr, o.cs, o.co, a.N, a.H * a.W, a.C, D.data_ptr(), _stream()), "sa_rowdot")


def sa_bwd(qkv: "TS", dp: int, C: int, dO: "TS", dopack: torch.Tensor, gpack: torch.Tensor, fpack: torch.Tensor, lse: torch.Tensor,
           D: torch.Tensor, dqkv: "TS"):
    """the gradient of the whole QKV buffer (dF | dG | dH slices) from dO = dL/dO, with the weights recomputed from lse"""
    assert qkv.bf16 and dO.bf16 and dqkv.bf16 and qkv.co == 0 and dqkv.co == 0 and dqkv.cs == qkv.cs and dO.C == C and dO.P == qkv.P
    rows, N = qkv.N * sa_rows(qkv.H * qkv.W), qkv.H * qkv.W
    assert lse.dtype == torch.float32 and D.dtype == torch.float32 and lse.numel() >= rows and D.numel() >= rows        # [B][64 ceil(N / 64)]
    assert dopack.numel() >= qkv.N * sa_pack_elems(N, C) and min(gpack.numel(), fpack.numel()) >= qkv.N * sa_pack_elems(N, dp)
    check(lib.unet_sa_bwd_bf16(qkv.ptr, qkv.cs, dp, C, qkv.N, qkv.H * qkv.W, dO.ptr, dO.cs, dO.co, dopack.data_ptr(), gpack.data_ptr(),
                               fpack.data_ptr(), lse.data_ptr(), D.data_ptr(), dqkv.ptr, _stream()), "sa_bwd")


def cast_slice(x: "TS", y: "TS"):
    """fp32 slice -> bf16 slice (same geometry)"""
    assert not x.bf16 and y.bf16 and x.P == y.P and x.C == y.C
    check(lib.unet_cast_slice_bf16(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.P, x.C, _stream()), "cast_slice")


def _conv_desc(x: TS, wp: torch.Tensor, y: TS, ks: int, stride: int, kind: int, bias=None, res: Optional[TS] = None,
               mask: Optional[TS] = None, relu=False, colsum=None, colsumsq=None) -> ConvDesc:
    d = ConvDesc()
    d.x, d.x_cs, d.x_co = x.ptr, x.cs, x.co
    d.wp = wp.data_ptr()
    d.bias = _p(bias)
    if res is not None:
        d.res, d.res_cs, d.res_co = res.ptr, res.cs, res.co
    flags = L.CONV_RELU if relu else 0
    if mask is not None:
        d.mask, d.mask_cs, d.mask_co = mask.ptr, mask.cs, mask.co
        flags |= L.CONV_MASK
    d.y, d.y_cs, d.y_co = y.ptr, y.cs, y.co
    d.N, d.IH, d.IW, d.Cin = x.N, x.H, x.W, x.C
    d.OH, d.OW, d.Cout = y.H, y.W, y.C
    d.ks, d.stride, d.kind, d.flags = ks, stride, kind, flags
    d.colsum, d.colsumsq = _p(colsum), _p(colsumsq)
    d.tuning = _tuning_ptr()
    if x.bf16:
        assert wp.dtype == torch.bfloat16 and (res is None or res.bf16) and (mask is None or mask.bf16)
        d.dtype, d.y_f32 = L.BF16, int(not y.bf16)
    else:
        assert wp.dtype == torch.float32 and not y.bf16
    return d


class ConvProbe:
    """bench.py: brackets every conv launch of one kernel instantiation with events on the launch stream and
    sums the ALGORITHMIC FLOPs (2 * pixels * Cin * Cout * k*k, no padding) of those launches."""

    def __init__(self, variant: int):
        self.variant = variant
        self.events = []
        self.flops = 0.0
        self.bytes = 0.0          # ALGORITHMIC bytes of the probed launches: every operand tensor once in, the result once out
        self.detail = []          # (what, N, H, W, Cin, Cout, ks, stride, flops) per probed launch
        # a stream whose work must not run next to a probed launch (the side stream of the weight gradients): the launch stream waits for
        # it in front of the first event, so that the events time THIS kernel alone, not this kernel sharing the chip with another one
        self.exclusive_of = None

    def summary(self):
        ms = [a.elapsed_time(b) for a, b in self.events]
        return {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / max(1, len(ms)), "flops": self.flops, "bytes": self.bytes}


CONV_PROBE: Optional[ConvProbe] = None


# Scratch for split-K conv launches (unet_conv_desc.splitk_ws): one fp32 buffer per device, grown on demand, reused by every launch
# (stream ordered).  It reaches its final size during the eager warm-up steps, so a captured step sees a static address.
_SPLITK_WS = {}
_SPLITK_RETIRED = []      # outgrown buffers stay alive: a captured hipGraph may hold their address in its split launches


def _splitk_ws(d: ConvDesc):
    need = int(lib.unet_conv2d_splitk_workspace(C.byref(d)))
    if need == 0:
        d.splitk_ws, d.splitk_ws_floats = None, 0
        return
    dev = _device_index()
    buf = _SPLITK_WS.get(dev)
    if buf is None or buf.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise L.UnetHipError("split-K workspace would have to grow inside a hipGraph capture: run the eager warm-up at this geometry first")
        if buf is not None:
            _SPLITK_RETIRED.append(buf)
        buf = _SPLITK_WS[dev] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=f"cuda:{dev}")
    d.splitk_ws, d.splitk_ws_floats = buf.data_ptr(), buf.numel()


def _launch_conv_part(d: ConvDesc, what: str, alg_flops: float):
    _splitk_ws(d)
    pr = CONV_PROBE
    if pr is not None and lib.unet_conv2d_variant(C.byref(d)) % 1000000 == pr.variant:      # (split-K launches of the instantiation included)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if pr.exclusive_of is not None:
            torch.cuda.current_stream().wait_stream(pr.exclusive_of)
        e0.record()
        check(lib.unet_conv2d(C.byref(d), _stream()), what)
        e1.record()
        pr.events.append((e0, e1))
        pr.flops += alg_flops
        # x + y (+ residual, + mask) once each at the storage width, the packed filter image once, the bias vector
        es = 2 if d.dtype == L.BF16 else 4
        co = d.cout_count or d.Cout
        pix_in, pix_out = d.N * d.IH * d.IW, d.N * d.OH * d.OW
        pr.bytes += (es * (pix_in * d.Cin + pix_out * co * (1 + (1 if d.res else 0) + (1 if d.mask else 0)))
                     + (4 - es) * pix_out * co * d.y_f32 + es * d.ks * d.ks * d.Cin * co)
        pr.detail.append((what, d.N, d.OH, d.OW, d.Cin, d.cout_count or d.Cout, d.ks, d.stride, alg_flops))
        return
    check(lib.unet_conv2d(C.byref(d), _stream()), what)


def _launch_conv(d: ConvDesc, what: str, alg_flops: float):
    """A produced-channel count of k*128 + r with 0 < r <= 64 would run its last 128-wide channel block at most half full
    (and re-stage the input tile for it): issue the k*128 part and the r-wide part as two launches, the second with a 64- or
    32-wide channel block.  Column sums (rows depend on the tile geometry of ONE launch) keep the single launch."""
    Co = d.Cout
    r = Co % 128
    if Co > 128 and 0 < r <= 64 and not d.colsum:
        for b, n in ((0, Co - r), (Co - r, r)):
            d.cout_begin, d.cout_count = b, n
            _launch_conv_part(d, what, alg_flops * n / Co)
        d.cout_begin = d.cout_count = 0
        return
    _launch_conv_part(d, what, alg_flops)


def conv2d(x: TS, wp: torch.Tensor, y: TS, ks: int, stride: int = 1, bias=None, res=None, mask=None, relu=False,
           colsum=None, colsumsq=None, wp_img_stride: int = 0):
    """wp_img_stride > 0: image n of the batch uses the packed filter image at wp + n * wp_img_stride floats"""
    d = _conv_desc(x, wp, y, ks, stride, L.CONV_FWD, bias, res, mask, relu, colsum, colsumsq)
    d.wp_img_stride = wp_img_stride
    _launch_conv(d, "conv2d", 2.0 * y.P * x.C * y.C * ks * ks)


def _ps_desc(x: TS, wp: torch.Tensor, X: TS, bias, relu: bool) -> ConvDesc:
    """1x1 conv of 4 * X.C channels whose epilogue stores PixelShuffle(2)(act(conv(x))) into the slice X of a [N, 2H, 2W] buffer"""
    assert X.H == 2 * x.H and X.W == 2 * x.W and X.N == x.N and x.bf16 == X.bf16
    d = ConvDesc()
    d.x, d.x_cs, d.x_co = x.ptr, x.cs, x.co
    d.wp, d.bias = wp.data_ptr(), _p(bias)
    d.y, d.y_cs, d.y_co = X.ptr, X.cs, X.co
    d.N, d.IH, d.IW, d.Cin = x.N, x.H, x.W, x.C
    d.OH, d.OW, d.Cout = x.H, x.W, 4 * X.C
    d.ks, d.stride, d.kind, d.flags = 1, 1, L.CONV_FWD, (L.CONV_RELU if relu else 0)
    d.dtype = L.BF16 if x.bf16 else L.F32
    d.tuning = _tuning_ptr()
    d.pixel_shuffle = 1
    return d


def conv1x1_shuffle_applies(x: TS, X: TS) -> bool:
    """does the library take conv1x1 -> act -> PixelShuffle(2) as ONE launch for this geometry (unet_conv2d_variant == 8)?"""
    d = _ps_desc(x, torch.empty(0), X, None, True)
    d.wp = 16          # (any aligned non-null address: the query only plans)
    return int(lib.unet_conv2d_variant(C.byref(d))) == 8


def conv1x1_shuffle_tail_ok(X: TS, tail: TS, at: int) -> bool:
    """can conv1x1_shuffle append `tail`'s channels at channel `at` of X's buffer (unet_conv_desc.ps_tail: whole quads at quad-aligned places)?"""
    return (tail.bf16 == X.bf16 and (tail.N, tail.H, tail.W) == (X.N, X.H, X.W) and tail.cs % 4 == 0 and tail.co % 4 == 0 and at % 4 == 0
            and tail.co + rup4(tail.C) <= tail.cs and at >= X.co + X.C and at + rup4(tail.C) <= X.cs)


def conv1x1_shuffle(x: TS, wp_ps: torch.Tensor, X: TS, bias=None, relu=True, tail: Optional[TS] = None, tail_at: int = 0):
    """X = PixelShuffle(2)(act(conv1x1(x) + bias)); wp_ps is the mode-2 packed image (columns in pixel-shuffle order).
    tail: an NHWC slice of X's geometry copied to channels [tail_at, tail_at + C) of X's BUFFER by the same launch (the network input behind
    the up-sampled channels of the final concat)"""
    d = _ps_desc(x, wp_ps, X, bias, relu)
    if tail is not None:
        assert conv1x1_shuffle_tail_ok(X, tail, tail_at)
        d.ps_tail, d.ps_tail_cs, d.ps_tail_co, d.ps_tail_c, d.ps_tail_at = tail.ptr, tail.cs, tail.co, tail.C, tail_at
    check(lib.unet_conv2d(C.byref(d), _stream()), "conv1x1_shuffle")


def shuffle_bwd_xmask(dX: TS, X: TS, dyc: TS):
    """dyc[h,w,4c+2i+j] = X[2h+i,2w+j,c] > 0 ? dX[2h+i,2w+j,c] : 0 (the adjoint of conv1x1_shuffle's store incl. its ReLU)"""
    assert dyc.C == 4 * dX.C and dX.H == 2 * dyc.H and dX.W == 2 * dyc.W and X.C == dX.C
    check(_fn("shuffle_bwd_xmask", dX)(dX.ptr, dX.cs, dX.co, X.ptr, X.cs, X.co, dyc.ptr, dyc.cs, dyc.co, dyc.N, dyc.H, dyc.W, dX.C, _stream()),
          "shuffle_bwd_xmask")


def conv2d_variant(x: TS, wp: torch.Tensor, y: TS, ks: int, stride: int = 1, kind: int = 0) -> int:
    """id of the kernel instantiation the planner picks for this launch (unet_conv2d_variant; scripts/layer_table.py, tests)"""
    d = _conv_desc(x, wp, y, ks, stride, kind)
    _splitk_ws(d)
    return int(lib.unet_conv2d_variant(C.byref(d)))


def conv2d_dgrad(dy: TS, wp_dgrad: torch.Tensor, dx: TS, ks: int, stride: int = 1, res=None, mask=None, colsum=None):
    """dx = conv^T(dy); optional residual add, ReLU-backward mask, column sums of the result."""
    d = _conv_desc(dy, wp_dgrad, dx, ks, stride, L.CONV_DGRAD, None, res, mask, False, colsum, None)
    _launch_conv(d, "conv2d_dgrad", 2.0 * dy.P * dy.C * dx.C * ks * ks)


def conv_colsum_rows(x: TS, wp, y: TS, ks, stride, kind) -> int:
    d = _conv_desc(x, wp, y, ks, stride, kind)
    r = lib.unet_conv2d_colsum_rows(C.byref(d))
    if r < 0:
        check(r, "conv2d_colsum_rows")
    return r


def _wgrad_desc(x: TS, dy: TS, dw, dbias, ks, stride, ws, accumulate) -> WgradDesc:
    d = WgradDesc()
    d.x, d.x_cs, d.x_co = x.ptr, x.cs, x.co
    d.dy, d.dy_cs, d.dy_co = dy.ptr, dy.cs, dy.co
    d.dw, d.dbias = _p(dw), _p(dbias)
    d.N, d.IH, d.IW, d.Cin = x.N, x.H, x.W, x.C
    d.OH, d.OW, d.Cout = dy.H, dy.W, dy.C
    d.ks, d.stride = ks, stride
    d.workspace = _p(ws)
    d.workspace_floats = 0 if ws is None else ws.numel()
    d.accumulate = int(accumulate)
    d.tuning = _tuning_ptr()
    assert x.bf16 == dy.bf16
    d.dtype = L.BF16 if x.bf16 else L.F32
    return d


def wgrad_workspace(x: TS, dy: TS, ks, stride, with_bias=False) -> int:
    d = _wgrad_desc(x, dy, torch.empty(0), None, ks, stride, None, 0)
    d.dw = 1  # non-null placeholder for validation
    return int(lib.unet_conv2d_wgrad_workspace(C.byref(d)))


def conv2d_wgrad(x: TS, dy: TS, dw: torch.Tensor, ks: int, stride: int, ws: torch.Tensor, dbias=None, accumulate=False):
    d = _wgrad_desc(x, dy, dw, dbias, ks, stride, ws, accumulate)
    check(lib.unet_conv2d_wgrad(C.byref(d), _stream()), "conv2d_wgrad")


# ------------------------------------------------------------------ batch norm

def bn_stats_rows(P: int) -> int:
    return lib.unet_bn_stats_rows(P)


def bn_stats(x: TS, partial: torch.Tensor):
    check(_fn("bn_stats", x)(x.ptr, x.cs, x.co, x.P, x.C, partial.data_ptr(), _stream()), "bn_stats")


def bn_finalize(psum, psumsq, rows, count, C_, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, sinvstd, tracked=None):
    assert tracked is None or tracked.dtype == torch.int64
    check(lib.unet_bn_finalize(_p(psum), _p(psumsq), rows, count, C_, _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                               _p(scale), _p(shift), _p(smean), _p(sinvstd), _p(tracked), _stream()), "bn_finalize")


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps, scale, shift):
    check(lib.unet_bn_eval_coeffs(_p(gamma), _p(beta), _p(rmean), _p(rvar), eps, rmean.numel(), _p(scale), _p(shift), _stream()),
          "bn_eval_coeffs")


def affine_act(x: TS, y: TS, scale=None, shift=None, x2: Optional[TS] = None, scale2=None, shift2=None, relu=False):
    check(_fn("affine_act", x)(x.ptr, x.cs, x.co, _p(scale), _p(shift),
                              None if x2 is None else x2.ptr, 0 if x2 is None else x2.cs, 0 if x2 is None else x2.co,
                              _p(scale2), _p(shift2), y.ptr, y.cs, y.co, x.P, x.C, int(relu), _stream()), "affine_act")


def bn_bwd_reduce(dout: TS, out: Optional[TS], x: TS, mean, invstd, partial):
    check(_fn("bn_bwd_reduce", x)(dout.ptr, dout.cs, dout.co, None if out is None else out.ptr, 0 if out is None else out.cs,
                                 0 if out is None else out.co, x.ptr, x.cs, x.co, _p(mean), _p(invstd), x.P, x.C,
                                 partial.data_ptr(), _stream()), "bn_bwd_reduce")


def bn_bwd_finalize(partial, rows, count, C_, dgamma, dbeta, c1, c2):
    check(lib.unet_bn_bwd_finalize(_p(partial), rows, count, C_, _p(dgamma), _p(dbeta), _p(c1), _p(c2), _stream()), "bn_bwd_finalize")


def bn_bwd_apply(dout: TS, out: Optional[TS], x: TS, mean, invstd, gamma, c1, c2, dx: TS, gout: Optional[TS] = None,
                 g_accumulate=False):
    check(_fn("bn_bwd_apply", x)(dout.ptr, dout.cs, dout.co, None if out is None else out.ptr, 0 if out is None else out.cs,
                                0 if out is None else out.co, x.ptr, x.cs, x.co, _p(mean), _p(invstd), _p(gamma), _p(c1), _p(c2),
                                dx.ptr, dx.cs, dx.co, None if gout is None else gout.ptr, 0 if gout is None else gout.cs,
                                0 if gout is None else gout.co, int(g_accumulate), x.P, x.C, _stream()), "bn_bwd_apply")


# ------------------------------------------------------------------ pooling

def maxpool(x: TS, y: TS, idx: Optional[torch.Tensor]):
    check(_fn("maxpool3x3s2", x)(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, _p(idx), x.N, x.H, x.W, x.C, y.H, y.W, _stream()), "maxpool")


def maxpool_bwd(dy: TS, idx: torch.Tensor, dx: TS, accumulate=False):
    check(_fn("maxpool3x3s2_bwd", dy)(dy.ptr, dy.cs, dy.co, idx.data_ptr(), dx.ptr, dx.cs, dx.co, dx.N, dx.H, dx.W, dx.C, dy.H, dy.W,
                                    int(accumulate), _stream()), "maxpool_bwd")


def avgpool(x: TS, y: TS):
    check(_fn("avgpool2_ceil", x)(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.N, x.H, x.W, x.C, y.H, y.W, _stream()), "avgpool")


def avgpool_bwd(dy: TS, dx: TS, accumulate=False):
    check(_fn("avgpool2_ceil_bwd", dy)(dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, dx.N, dx.H, dx.W, dx.C, dy.H, dy.W, int(accumulate),
                                     _stream()), "avgpool_bwd")


# ------------------------------------------------------------------ decoder data movement

def shuffle_blur(yc: TS, X: TS, blur: bool):
    assert yc.C == 4 * X.C and X.H == 2 * yc.H and X.W == 2 * yc.W
    check(_fn("shuffle_blur", yc)(yc.ptr, yc.cs, yc.co, X.ptr, X.cs, X.co, yc.N, yc.H, yc.W, X.C, int(blur), _stream()), "shuffle_blur")


def shuffle_blur_bwd(dX: TS, yc: TS, dyc: TS, blur: bool):
    assert yc.C == 4 * dX.C and dX.H == 2 * yc.H and dX.W == 2 * yc.W
    check(_fn("shuffle_blur_bwd", dX)(dX.ptr, dX.cs, dX.co, yc.ptr, yc.cs, yc.co, dyc.ptr, dyc.cs, dyc.co, yc.N, yc.H, yc.W, dX.C,
                                    int(blur), _stream()), "shuffle_blur_bwd")


def resize_nearest(x: TS, y: TS):
    assert x.bf16 == y.bf16
    check(_fn("resize_nearest", x)(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.N, x.H, x.W, y.H, y.W, x.C, _stream()), "resize_nearest")


def resize_nearest_bwd(dy: TS, dx: TS):
    assert dy.bf16 == dx.bf16
    check(_fn("resize_nearest_bwd", dy)(dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, dx.N, dx.H, dx.W, dy.H, dy.W, dx.C, _stream()),
          "resize_nearest_bwd")


def nchw_to_nhwc(x: torch.Tensor, y: TS, at: Optional[int] = None):
    """NCHW fp32 -> the channels of the NHWC slice `y`; with `at`, into channels [at, at + C) of y's buffer instead -- the one writer
    that may start at ANY channel (scalar stores): the network-input half of the final concat sits behind an up-sampling path whose
    width need not be a multiple of the vector width (xresnet34_deep: 102)."""
    N, C_, H, W = x.shape
    if at is None:
        assert x.is_contiguous() and y.C == C_
        check(_fn("nchw_to_nhwc", y)(x.data_ptr(), y.ptr, y.cs, y.co, N, C_, H, W, _stream()), "nchw_to_nhwc")
    else:
        assert x.is_contiguous() and y.co == 0 and at + C_ <= y.cs
        check(_fn("nchw_to_nhwc", y)(x.data_ptr(), y.ptr, y.cs, at, N, C_, H, W, _stream()), "nchw_to_nhwc")


def nhwc_to_nchw(x: TS, y: torch.Tensor):
    _need_f32("nhwc_to_nchw", x)
    assert y.is_contiguous()
    check(lib.unet_nhwc_to_nchw(x.ptr, x.cs, x.co, y.data_ptr(), x.N, x.C, x.H, x.W, _stream()), "nhwc_to_nchw")


def copy_slice(x: TS, y: TS, accumulate=False):
    check(_fn("copy_slice", x)(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.P, x.C, int(accumulate), _stream()), "copy_slice")


def relu_mask(g: TS, ref: TS, y: TS):
    assert g.bf16 == ref.bf16 == y.bf16
    check(_fn("relu_mask", g)(g.ptr, g.cs, g.co, ref.ptr, ref.cs, ref.co, y.ptr, y.cs, y.co, g.P, g.C, _stream()), "relu_mask")


def colsum_workspace(P, C_) -> int:
    return int(lib.unet_colsum_workspace(P, C_))


def colsum(x: TS, out: torch.Tensor, ws: torch.Tensor):
    _need_f32("colsum", x)
    assert ws.numel() >= colsum_workspace(x.P, x.C)
    check(lib.unet_colsum(x.ptr, x.cs, x.co, x.P, x.C, out.data_ptr(), ws.data_ptr(), _stream()), "colsum")


def dot(x: TS, y: TS, out: torch.Tensor, ws: torch.Tensor):
    """out[0] = sum(x * y) over all pixels and channels (fp32 accumulation)"""
    assert x.bf16 == y.bf16
    assert ws.numel() >= colsum_workspace(x.P, x.C)
    check(_fn("dot", x)(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.P, x.C, out.data_ptr(), ws.data_ptr(), _stream()), "dot")


# ------------------------------------------------------------------ loss

def ce_workspace(P) -> int:
    return int(lib.unet_ce_workspace(P))


def ce_fwd(z: TS, target: torch.Tensor, weight, loss, denom, ws):
    _need_f32("ce_fwd", z)
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == z.P
    check(lib.unet_ce_fwd(z.ptr, z.cs, z.co, target.data_ptr(), _p(weight), z.P, z.C, loss.data_ptr(), denom.data_ptr(),
                          ws.data_ptr(), _stream()), "ce_fwd")


def ce_fwd_parts(z: TS, target: torch.Tensor, weight, numden: torch.Tensor, ws):
    """numden[0] = sum w[y] * nll, numden[1] = sum w[y] over this rank's pixels (tile-DDP: all-reduced before the division)"""
    _need_f32("ce_fwd_parts", z)
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == z.P and numden.numel() >= 2
    check(lib.unet_ce_fwd_parts(z.ptr, z.cs, z.co, target.data_ptr(), _p(weight), z.P, z.C, numden.data_ptr(), ws.data_ptr(), _stream()),
          "ce_fwd_parts")


def ce_bwd(z: TS, target, weight, denom, gscale: float, dz: TS):
    check(_fn("ce_bwd", dz)(z.ptr, z.cs, z.co, target.data_ptr(), _p(weight), z.P, z.C, denom.data_ptr(), gscale, dz.ptr, dz.cs,
                          dz.co, _stream()), "ce_bwd")


def focal_fwd(z: TS, target: torch.Tensor, weight, gamma: float, loss, ws):
    """loss[0] = mean over all pixels of (1 - exp(-ce))^gamma * ce, ce = w[y] * nll (fastai FocalLossFlat(gamma, axis=1))"""
    _need_f32("focal_fwd", z)
    assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == z.P
    check(lib.unet_focal_fwd(z.ptr, z.cs, z.co, target.data_ptr(), _p(weight), z.P, z.C, float(gamma), loss.data_ptr(), ws.data_ptr(), _stream()),
          "focal_fwd")


def focal_bwd(z: TS, target, weight, gamma: float, gscale: float, dz: TS):
    check(_fn("focal_bwd", dz)(z.ptr, z.cs, z.co, target.data_ptr(), _p(weight), z.P, z.C, float(gamma), gscale, dz.ptr, dz.cs, dz.co, _stream()),
          "focal_bwd")


REG_KINDS = {"mse": 0, "l1": 1, "smoothl1": 2}


def regloss_fwd(z: TS, target: torch.Tensor, kind: str, beta: float, loss, ws):
    _need_f32("regloss_fwd", z)
    assert target.dtype == torch.float32 and target.is_contiguous() and target.numel() == z.P and z.C == 1
    check(lib.unet_regloss_fwd(z.ptr, z.cs, z.co, target.data_ptr(), z.P, REG_KINDS[kind], float(beta), loss.data_ptr(), ws.data_ptr(),
                               _stream()), "regloss_fwd")


def regloss_bwd(z: TS, target: torch.Tensor, kind: str, beta: float, gscale: float, dz: TS):
    _need_f32("regloss_bwd", z, dz)
    check(lib.unet_regloss_bwd(z.ptr, z.cs, z.co, target.data_ptr(), z.P, REG_KINDS[kind], float(beta), float(gscale), dz.ptr, dz.cs,
                               dz.co, _stream()), "regloss_bwd")


def softmax_argmax(z: TS, probs: Optional[torch.Tensor], amax: Optional[torch.Tensor]):
    _need_f32("softmax_argmax", z)
    check(lib.unet_softmax_argmax(z.ptr, z.cs, z.co, z.N, z.H, z.W, z.C, _p(probs), _p(amax), _stream()), "softmax_argmax")


# ------------------------------------------------------------------ optimiser

def adam_step(p, g, m, v, code, lrs, mom, sqr_mom, eps, wd, step, grad_scale=1.0):
    arr = (C.c_float * 4)(*([float(l) for l in lrs] + [0.0] * (4 - len(lrs))))
    check(lib.unet_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), code.data_ptr(), p.numel(), arr,
                             float(mom), float(sqr_mom), float(eps), float(wd), int(step), float(grad_scale), _stream()), "adam_step")


def mosaic_accumulate(probs: torch.Tensor, mosaic: torch.Tensor, count: torch.Tensor, y0: int, x0: int):
    Cc, th, tw = probs.shape
    _, MH, MW = mosaic.shape
    check(lib.unet_mosaic_accumulate(probs.data_ptr(), Cc, th, tw, mosaic.data_ptr(), count.data_ptr(), MH, MW, y0, x0, _stream()),
          "mosaic_accumulate")


def mosaic_finalize(mosaic: torch.Tensor, count: torch.Tensor, amax: Optional[torch.Tensor]):
    Cc, MH, MW = mosaic.shape
    check(lib.unet_mosaic_finalize(mosaic.data_ptr(), count.data_ptr(), Cc, MH, MW, _p(amax), _stream()), "mosaic_finalize")


# ------------------------------------------------------------------ sliding-window predict over an integer raster (csrc/raster.hip)

RASTER_TYPES = {torch.uint8: 0, torch.uint16: 1, torch.int16: 2, torch.int32: 3, torch.float32: 4}


class WindowSource:
    """Band-sequential integer (or float) samples on the device that windows are cut from: ONE raster [C, H, W] (``src_stride`` 0) or a
    batch of staged tiles [n, C, h, w].  Sample (source s, band c, row y, column x) = data[s, c, y, x]."""

    def __init__(self, data: torch.Tensor, div255_twice: bool = False):
        assert data.is_cuda and data.is_contiguous() and data.dim() in (3, 4) and data.dtype in RASTER_TYPES, (data.shape, data.dtype)
        self.data, self.rtype, self.div2 = data, RASTER_TYPES[data.dtype], int(bool(div255_twice))
        self.C, self.H, self.W = data.shape[-3:]
        self.src_stride = 0 if data.dim() == 3 else self.C * self.H * self.W


def window_table(rows, device) -> torch.Tensor:
    """int32 [n, 4] device table of (y0, x0, source index, 0)"""
    t = torch.zeros((len(rows), 4), dtype=torch.int32)
    if len(rows):
        a = torch.as_tensor(rows, dtype=torch.int32)
        t[:, :a.shape[1]] = a
    return t.to(device)


@dataclass
class WindowBatch:
    """windows [first, first + n) of a device window table over a WindowSource: an input batch of the network that is never
    materialised as an NCHW fp32 tensor -- HipDynamicUnet writes it straight into its NHWC input buffers (window_gather)"""
    src: WindowSource
    table: torch.Tensor
    first: int
    n: int
    th: int
    tw: int

    def write(self, dst_buf: torch.Tensor, at: int):
        window_gather(self.src, self.table, self.first, self.n, self.th, self.tw, dst_buf, at)


def raster_nodata_zero(src: WindowSource, nodata: float):
    assert src.data.dim() == 3
    check(lib.unet_raster_nodata_zero(src.data.data_ptr(), src.rtype, src.C, src.H * src.W, float(nodata), _stream()), "raster_nodata_zero")


def window_nonzero(src: WindowSource, table: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    """int64 [n]: non-zero samples of every window (all bands)"""
    n = table.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=table.device)
    if n:
        check(lib.unet_window_nonzero(src.data.data_ptr(), src.rtype, src.C, src.H * src.W, src.W, table.data_ptr(), n, th, tw,
                                      out.data_ptr(), _stream()), "window_nonzero")
    return out


def window_gather(src: WindowSource, table: torch.Tensor, first: int, n: int, th: int, tw: int, dst_buf: torch.Tensor, at: int):
    """windows [first, first + n) of the table -> channels [at, at + C) of the NHWC buffer dst_buf [>= n, th, tw, cs], scaled"""
    assert dst_buf.dim() == 4 and dst_buf.shape[0] >= n and tuple(dst_buf.shape[1:3]) == (th, tw) and dst_buf.is_contiguous()
    assert 0 <= first and first + n <= table.shape[0]
    dt = L.BF16 if dst_buf.dtype == torch.bfloat16 else L.F32
    check(lib.unet_window_gather(src.data.data_ptr(), src.rtype, src.C, src.src_stride, src.H * src.W, src.W,
                                 table.data_ptr() + 16 * first, n, th, tw, src.div2, dst_buf.data_ptr(), dst_buf.shape[3], at, dt,
                                 _stream()), "window_gather")


# ------------------------------------------------------------------ training feed (csrc/raster.hip: unet_tiles_stage / unet_mask_stage)

def _flip_bits(flips, n: int, at: int):
    """(hflip, vflip) bit masks of images [at, at + n) from a sequence of (h, v) pairs (None: nothing flipped)"""
    h = v = 0
    if flips is not None:
        for j in range(n):
            fh, fv = flips[at + j]
            h |= int(bool(fh)) << j
            v |= int(bool(fv)) << j
    return h, v


def tiles_stage(src: torch.Tensor, div255_twice: bool, out: torch.Tensor, flips=None):
    """staged integer tiles [n, C, H, W] (device) -> fp32 NCHW batch `out`, scaled like learner.scale_input (bit-equal); flips[j] = (h, v)"""
    assert src.is_cuda and src.is_contiguous() and src.dim() == 4 and src.dtype in RASTER_TYPES, (src.shape, src.dtype)
    assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == tuple(src.shape)
    n, Cb, H, W = src.shape
    per = Cb * H * W
    for at in range(0, n, 64):
        m = min(64, n - at)
        hb, vb = _flip_bits(flips, m, at)
        check(lib.unet_tiles_stage(src.data_ptr() + at * per * src.element_size(), RASTER_TYPES[src.dtype], m, Cb, H, W, int(bool(div255_twice)),
                                   hb, vb, out.data_ptr() + at * per * 4, _stream()), "tiles_stage")


def mask_stage(src: torch.Tensor, out: torch.Tensor, flips=None):
    """staged masks [n, H, W] (device, any raster sample type) -> `out` int64 (classification) or float32 (regression targets)"""
    assert src.is_cuda and src.is_contiguous() and src.dim() == 3 and src.dtype in RASTER_TYPES, (src.shape, src.dtype)
    assert out.is_cuda and out.is_contiguous() and out.dtype in (torch.int64, torch.float32) and tuple(out.shape) == tuple(src.shape)
    n, H, W = src.shape
    for at in range(0, n, 64):
        m = min(64, n - at)
        hb, vb = _flip_bits(flips, m, at)
        check(lib.unet_mask_stage(src.data_ptr() + at * H * W * src.element_size(), RASTER_TYPES[src.dtype], m, H, W, hb, vb,
                                  out.data_ptr() + at * H * W * out.element_size(), int(out.dtype == torch.float32), _stream()), "mask_stage")


def dice_counts(pred: torch.Tensor, targ: torch.Tensor, n_cls: int, counts: torch.Tensor):
    """counts int64 [3, n_cls] += (intersection, predicted, target) pixel counts per class of this batch (DiceMulti)"""
    assert pred.dtype == torch.int64 and targ.dtype == torch.int64 and pred.is_contiguous() and targ.is_contiguous()
    assert pred.numel() == targ.numel() and counts.dtype == torch.int64 and counts.numel() == 3 * n_cls and counts.is_contiguous()
    check(lib.unet_dice_counts(pred.data_ptr(), targ.data_ptr(), pred.numel(), n_cls, counts.data_ptr(), _stream()), "dice_counts")


def mosaic_accumulate_windows(z: TS, table: torch.Tensor, first: int, n: int, origin, mosaic: torch.Tensor, count: torch.Tensor,
                              row_lo: int, row_hi: int, raw: bool = False):
    """softmax (or, raw=True, the values themselves) of the fp32 NHWC logits of windows [first, first + n) added into mosaic / count"""
    _need_f32("mosaic_accumulate_windows", z)
    Cc, MH, MW = mosaic.shape
    assert z.N >= n and Cc == z.C and count.shape == (MH, MW) and count.dtype == torch.int32 and mosaic.dtype == torch.float32
    check(lib.unet_mosaic_accumulate_windows(z.ptr, z.cs, z.co, z.C, z.H, z.W, table.data_ptr() + 16 * first, n, int(origin[0]),
                                             int(origin[1]), int(raw), mosaic.data_ptr(), count.data_ptr(), MH, MW, int(row_lo), int(row_hi),
                                             _stream()), "mosaic_accumulate_windows")


def mosaic_finalize_rows(mosaic: torch.Tensor, count: torch.Tensor, row0: int, nrows: int, amax: Optional[torch.Tensor], fill=None):
    Cc, MH, MW = mosaic.shape
    assert amax is None or (amax.dtype == torch.uint8 and amax.numel() >= nrows * MW)
    fp = None if fill is None else C.byref(C.c_float(float(fill)))
    check(lib.unet_mosaic_finalize_rows(mosaic.data_ptr(), count.data_ptr(), Cc, MH, MW, row0, nrows, _p(amax),
                                        None if fp is None else C.cast(fp, L.c_float_p), _stream()), "mosaic_finalize_rows")
