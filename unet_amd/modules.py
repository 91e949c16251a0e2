"""Module tree of the MI355X DynamicUnet and its hand-written forward / backward programs.

The tree mirrors the object the reference builds at ``train.py:128-144``
(``create_body(xresnetNN)`` + stem swap + fastai ``DynamicUnet(blur=True,
blur_final=True, self_attention, norm_type=NormType, last_cross=True, bottle=False)``)
child for child, so that ``state_dict()`` keys (``layers.0.0.0.weight`` ...), the
splitter indexing ``m[0][:3] / m[0][3:] / m[1:]`` (``train.py:78-80``) and
``print(model)`` look like fastai's.  ``nn.Conv2d`` / ``nn.BatchNorm2d`` instances are
parameter holders only: the arithmetic runs in libunet_hip.so through
``hip_fwd`` / ``hip_bwd`` of the container blocks.  There is no autograd inside the
network and no CPU / eager fallback: calling ``forward`` of an inner block raises.

Gradient convention of the programs
    encoder blocks (conv -> BN -> ReLU):  ``hip_bwd`` receives dL/d(block output);
    decoder conv layers (conv + bias + ReLU, no norm): ``hip_bwd`` receives the gradient
    w.r.t. the PRE-activation, i.e. already multiplied by (output > 0): the ReLU backward
    of a layer is fused into the dgrad epilogue of its consumer (UNET_CONV_MASK).
"""
from __future__ import annotations

import collections
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import TS

BN_EPS = 1e-5
BN_MOM = 0.1


# --------------------------------------------------------------------------
# execution context: persistent device buffers (static addresses => graph-capturable)
# --------------------------------------------------------------------------

class Ctx:
    def __init__(self, device, act_dtype=torch.float32):
        self.device = device
        self.act_dtype = act_dtype          # storage type of activations / activation gradients / packed filters
        self.training = True
        self.need_grad = True
        self._acts: Dict[tuple, TS] = {}
        self._vecs: Dict[tuple, torch.Tensor] = {}
        self._pool: Dict[tuple, list] = {}
        self._pool_all: List[torch.Tensor] = []
        self._pool_key: Dict[int, tuple] = {}
        self._pool_live: set = set()
        self._pool_flat: Dict[int, torch.Tensor] = {}
        self._ws: Optional[torch.Tensor] = None
        self.saved: Dict[tuple, object] = {}
        # bumped whenever parameter memory is rewritten behind torch's back (HIP Adam step, all-reduce ...):
        # packed filter images older than this epoch are rebuilt
        self.weights_epoch = 0
        # bumped whenever BatchNorm running statistics move (a training-mode forward): cached eval coefficients and folded filter
        # images are older than this epoch
        self.bn_epoch = 0
        self.fold_bn = False                # this forward folds eval-mode BatchNorms into the conv in front of them
        # Weight gradients on a second stream (see _ConvExec.bwd_w): dL/dw of a layer is needed by nobody before the optimizer (or the
        # gradient all-reduce), dL/dx by the very next launch -- so wgrad + its split reduction leave the critical path and run next to
        # the input-gradient / BatchNorm-backward chain.  Large layers fill the chip either way; small grids (deep stages, small batches:
        # BASELINE configs[0]) and the HBM-bound reduce launches overlap.  UNET_WGRAD_STREAM=0 keeps everything on one stream.
        # Measured (scripts/ab_side.py, ab_side2.py, r04_j_graph_side.log).  The overlap pays where the step is GPU-bound and the two
        # chains are not both MFMA-saturated: bf16 storage at cfg2 size +3 %, fp32 cfg1 (batch 2 of 256^2) +7 %.  It costs ~5 HIP event / wait
        # calls per weight gradient on the host: the bf16 cfg1 step is HOST-bound and loses (3.97 ms without, 4.7-5.6 with).  The fp32 cfg2 step is
        # MFMA-bound in both chains and gains nothing (144.0 vs 144.0 tiles/s); moving only SOME launches is worse than either extreme (small
        # side-stream kernels starve behind chip-filling main-stream ones and the main stream then waits for them: 136-138).  So the switch is
        # per forward geometry, all weight gradients of the step or none: fp32 up to 2^20 input pixels per step, bf16 storage from 2^20 up.
        # UNET_WGRAD_STREAM = 0: never, all: always.
        mode = os.environ.get("UNET_WGRAD_STREAM", "1")
        self.wgrad_overlap = mode != "0"
        big = 1 << 62
        if mode == "all":
            self.wgrad_overlap_pixels, self.wgrad_overlap_min_pixels = big, 0
        elif act_dtype == torch.float32:
            self.wgrad_overlap_pixels, self.wgrad_overlap_min_pixels = 1 << 20, 0
        else:
            self.wgrad_overlap_pixels, self.wgrad_overlap_min_pixels = big, 1 << 20
        self.step_pixels = 0                # N * H * W of the current forward (set by HipDynamicUnet._hip_forward)
        self.main_stream = None             # the launch stream of the running backward program (fetched once per backward: torch.cuda.current_stream() is slow)
        self._side: Optional[torch.cuda.Stream] = None
        self._side_ws: Optional[torch.Tensor] = None
        self._side_dirty = False            # side-stream work launched since the last join
        self._side_tag: Dict[int, torch.cuda.Event] = {}     # pool buffer -> event behind its last side-stream reader
        self._side_pending = collections.deque()             # freed buffers withheld from the pool until that reader is done
        self.side_depth = int(os.environ.get("UNET_SIDE_DEPTH", "3"))     # how many such buffers may be withheld (= how far the main stream may run ahead)

    # activations are keyed by (owner id, tag, shape): allocated once per input geometry
    def act(self, owner, tag, N, H, W, C, zero=False, dtype=None) -> TS:
        key = (id(owner), tag, N, H, W, C)
        t = self._acts.get(key)
        if t is None:
            t = ops.new_act(N, H, W, C, self.device, zero=zero, dtype=self.act_dtype if dtype is None else dtype)
            self._acts[key] = t
        return t

    # Backward temporaries (activation gradients): every one is written by one backward block and read by the next, so they come
    # from a pool keyed by geometry and go back to it as soon as their consumer has been launched (launches are stream ordered:
    # the next user of the memory is queued behind the last reader).  LIFO free lists + a fixed program order = the same buffer
    # for the same role in every step, i.e. addresses stay static for hipGraph capture.  Padding lanes (cs != C) are zero at
    # allocation and every user of a (N, H, W, C) class writes only real channels or zeros there.
    def tmp(self, N, H, W, C, dtype=None) -> TS:
        dt = self.act_dtype if dtype is None else dtype
        cs = ops.rupv(C, dt)
        n = N * H * W * cs
        if cs != C:
            # padded channel count (rare: 3-band input, class count): exact-geometry class, so that the padding lanes stay zero
            key = ("exact", N, H, W, C, dt)
            free = self._pool.setdefault(key, [])
            flat = free.pop() if free else None
        else:
            # best fit by capacity over ALL geometries of this storage type: the pool's size is the peak of simultaneously live
            # bytes, not the sum of the per-shape peaks
            key = ("flat", dt)
            free = self._pool.setdefault(key, [])
            best = -1
            for i, b in enumerate(free):
                if b.numel() >= n and (best < 0 or b.numel() < free[best].numel()):
                    best = i
            flat = free.pop(best) if best >= 0 else None
        if flat is None:
            flat = torch.zeros(n, dtype=dt, device=self.device)
            self._pool_all.append(flat)
            self._pool_key[flat.data_ptr()] = key
        self._pool_live.add(flat.data_ptr())
        self._pool_flat[flat.data_ptr()] = flat
        return TS(flat[:n].view(N, H, W, cs), 0, C)

    def free(self, t: Optional[TS]):
        """return a ctx.tmp() buffer (any channel slice of it) to the pool; None and non-pool buffers are ignored"""
        if t is None:
            return
        ptr = t.buf.data_ptr()
        key = self._pool_key.get(ptr)
        if key is None:
            return
        if ptr not in self._pool_live:
            raise RuntimeError("backward temporary freed twice")
        self._pool_live.discard(ptr)
        ev = self._side_tag.pop(ptr, None)
        if ev is not None:
            # a weight-gradient launch on the side stream still reads this buffer: it stays out of the pool until the main stream has
            # been made to wait for that launch -- which happens `side_depth` frees later, when the launch is long done (no stall).
            # Program order decides which buffer comes back when: addresses stay the same from step to step
            self._side_pending.append((ptr, key, ev))
            while len(self._side_pending) > self.side_depth:
                self._side_release_oldest()
            return
        self._pool[key].append(self._pool_flat[ptr])

    # ---- the side stream of the weight gradients
    def side(self) -> Optional["torch.cuda.Stream"]:
        if not self.wgrad_overlap or self.device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def overlap_now(self) -> bool:
        """do the weight gradients (and the input-gradient filter images) of the current step run on the second stream?"""
        return self.wgrad_overlap and self.wgrad_overlap_min_pixels <= self.step_pixels <= self.wgrad_overlap_pixels

    def side_workspace(self, nfloats: int) -> torch.Tensor:
        """scratch of the side stream's launches (they are ordered among themselves; the main stream's ctx.workspace is not theirs)"""
        if self._side_ws is None or self._side_ws.numel() < nfloats:
            old = self._side_ws
            with torch.cuda.stream(self._side):
                self._side_ws = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=self.device)
            if old is not None:
                old.record_stream(self._side)          # (its last reader runs on the side stream: the allocator waits for it)
        return self._side_ws

    def side_reads(self, t: Optional[TS], ev) -> None:
        """`ev` (recorded on the side stream) lies behind the last side-stream reader of t's buffer"""
        if t is not None and t.buf.data_ptr() in self._pool_key:
            self._side_tag[t.buf.data_ptr()] = ev

    def _side_release_oldest(self):
        ptr, key, ev = self._side_pending.popleft()
        (self.main_stream or torch.cuda.current_stream()).wait_event(ev)
        self._pool[key].append(self._pool_flat[ptr])

    def side_join(self):
        """the current stream waits for every weight gradient launched so far (before the optimizer / a gradient bucket leaves)"""
        if self._side is not None and self._side_dirty:
            (self.main_stream or torch.cuda.current_stream()).wait_stream(self._side)
            self._side_dirty = False
        while self._side_pending:
            ptr, key, _ = self._side_pending.popleft()
            self._pool[key].append(self._pool_flat[ptr])
        self._side_tag.clear()

    def vec(self, owner, tag, n, dtype=torch.float32) -> torch.Tensor:
        key = (id(owner), tag, n, dtype)
        t = self._vecs.get(key)
        if t is None:
            t = torch.zeros(n, dtype=dtype, device=self.device)
            self._vecs[key] = t
        return t

    def workspace(self, nfloats: int) -> torch.Tensor:
        """One shared scratch buffer: every launch is stream ordered, so reuse is safe."""
        if self._ws is None or self._ws.numel() < nfloats:
            self._ws = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=self.device)
        return self._ws

    def bytes_allocated(self) -> int:
        n = sum(t.buf.numel() * t.buf.element_size() for t in self._acts.values()) + sum(v.numel() * v.element_size() for v in self._vecs.values())
        n += sum(b.numel() * b.element_size() for b in self._pool_all)
        return n + (0 if self._ws is None else self._ws.numel() * 4)


def _no_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} runs only inside HipDynamicUnet (HIP path); it has no eager forward")


# --------------------------------------------------------------------------
# leaf helpers: one convolution / one batch norm on the device
# --------------------------------------------------------------------------

# BatchNorm batch statistics from the producing conv's epilogue instead of a separate pass over y.  Parity-tested, but OFF: the
# per-wave partial rows (4096 for a 128x128 stage) make the finalize kernel slower than the pass it saves (step 122.5 -> 127.0 ms).
FUSE_BN_STATS = False
# conv1x1 + bias + ReLU + PixelShuffle(2) as one launch where there is no blur behind it (UNET_FUSE_SHUFFLE=0: the two-pass form, A/B)
FUSE_SHUFFLE = os.environ.get("UNET_FUSE_SHUFFLE", "1") != "0"
# ... and, with bf16 storage, the network-input half of the final concat appended by that launch (unet_conv_desc.ps_tail;
# UNET_FUSE_CONCAT_TAIL=0: a second pass, A/B)
FUSE_CONCAT_TAIL = os.environ.get("UNET_FUSE_CONCAT_TAIL", "1") != "0"


class _ConvExec:
    """Packed-weight cache + launches for one nn.Conv2d parameter holder."""

    def __init__(self, conv: nn.Conv2d):
        self.conv = conv
        self.ks = conv.kernel_size[0]
        self.stride = conv.stride[0]
        self.wp_f: Optional[torch.Tensor] = None
        self.wp_d: Optional[torch.Tensor] = None
        self.wp_s: Optional[torch.Tensor] = None      # forward image with its columns in pixel-shuffle order (ops.conv1x1_shuffle), built on demand
        self._ver_f = None
        self._ver_d = None
        self._ver_s = None
        self.ctx: Optional["Ctx"] = None   # set by HipDynamicUnet once the tree is built
        self.fold: Optional["_BNExec"] = None       # the BatchNorm that follows this conv (ConvLayer with norm), folded in eval mode
        # Channel gaps (bf16 storage of a concat whose first part is not a multiple of 8 channels wide: xresnet34_deep).  The tensors carry
        # `gap` dead channels at [at, at + gap) of the input and / or output dimension; the kernels see filters / a bias with zeros there.
        self.in_gap: Optional[Tuple[int, int]] = None      # (at, gap)
        self.out_gap: Optional[Tuple[int, int]] = None
        self._wpad = self._bpad = self._gwpad = self._gbpad = None
        self._ver_pad = None
        self._ws_sizes: Dict[tuple, int] = {}

    def set_gaps(self, in_gap=None, out_gap=None):
        self.in_gap = in_gap if in_gap and in_gap[1] else None
        self.out_gap = out_gap if out_gap and out_gap[1] else None
        self._wpad = None

    @property
    def gapped(self) -> bool:
        return self.in_gap is not None or self.out_gap is not None

    @staticmethod
    def _segs(n, gap):
        """[(logical slice, physical slice)] of a dimension of n logical channels"""
        if gap is None:
            return [(slice(0, n), slice(0, n))], n
        at, g = gap
        return [(slice(0, at), slice(0, at)), (slice(at, n), slice(at + g, n + g))], n + g

    def wsrc(self) -> torch.Tensor:
        """the filter the kernels pack: the parameter itself, or its copy with zero rows / columns for the gap channels"""
        w = self.conv.weight.data
        if not self.gapped:
            return w
        Cout, Cin = w.shape[:2]
        si, ci = self._segs(Cin, self.in_gap)
        so, co = self._segs(Cout, self.out_gap)
        if self._wpad is None:
            self._wpad = torch.zeros((co, ci) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
            self._gwpad = torch.zeros_like(self._wpad)
            if self.conv.bias is not None:
                self._bpad = torch.zeros(co, dtype=w.dtype, device=w.device)
                self._gbpad = torch.zeros_like(self._bpad)
            self._ver_pad = None
        ver = (self.conv.weight._version, w.data_ptr(), 0 if self.ctx is None else self.ctx.weights_epoch)
        if self._ver_pad != ver:
            for lo, po in so:
                for li, pi in si:
                    self._wpad[po, pi].copy_(w[lo, li])
                if self._bpad is not None:
                    self._bpad[po].copy_(self.conv.bias.data[lo])
            self._ver_pad = ver
        return self._wpad

    def bsrc(self) -> Optional[torch.Tensor]:
        b = self.conv.bias
        if b is None:
            return None
        if not self.gapped:
            return b.data
        self.wsrc()
        return self._bpad

    def fold_scale(self) -> Optional[torch.Tensor]:
        """per-output-channel factor of the forward image: the eval-mode BatchNorm scale when this forward folds it, else None"""
        if self.fold is None or self.ctx is None or not self.ctx.fold_bn:
            return None
        return self.fold.coeffs(self.ctx, None)[0]

    def packed(self, mode: int) -> torch.Tensor:
        ver = self.version()
        if mode == 0:
            if self.wp_f is None or self._ver_f != ver:
                self.ensure_buffers(False)
                ops.pack_jobs([(self.wsrc(), self.wp_f, 0, self.fold_scale())], self.wp_f.dtype == torch.bfloat16, self.wp_f.device)
                self._ver_f = ver
            return self.wp_f
        if mode == 2:
            dt = torch.float32 if self.ctx is None else self.ctx.act_dtype
            if self.wp_s is None or self.wp_s.dtype != dt or self._ver_s != ver:
                self.wp_s = ops.pack_weights(self.conv.weight.data, 2, self.wp_s, dtype=dt)
                self._ver_s = ver
            return self.wp_s
        if self.wp_d is None or self._ver_d != ver:
            self.ensure_buffers(True)
            ops.pack_jobs([(self.wsrc(), self.wp_d, 1, None)], self.wp_d.dtype == torch.bfloat16, self.wp_d.device)
            self._ver_d = ver
        return self.wp_d

    def version(self):
        w = self.conv.weight
        c = self.ctx
        folded = c is not None and c.fold_bn and self.fold is not None
        if not folded:
            return (w._version, w.data_ptr(), 0 if c is None else c.weights_epoch, -1)
        bn = self.fold.bn                # the folded scale follows the BatchNorm's tensors as _BNExec.coeffs does (a direct buffer update counts)
        return (w._version, w.data_ptr(), c.weights_epoch, c.bn_epoch, bn.weight._version, bn.running_var._version, bn.running_mean._version,
                bn.bias._version)

    def ensure_buffers(self, with_dgrad: bool):
        """the persistent packed-image buffers (HipDynamicUnet packs all of them in one launch: unet_pack_batch_run)"""
        dt = torch.float32 if self.ctx is None else self.ctx.act_dtype
        w = self.wsrc() if self.gapped else self.conv.weight
        Cout, Cin, ks, _ = w.shape
        size = ops.lib.unet_pack_weights_size_bf16 if dt == torch.bfloat16 else ops.lib.unet_pack_weights_size
        if self.wp_f is None or self.wp_f.dtype != dt:
            self.wp_f = torch.empty(size(Cout, Cin, ks, 0), dtype=dt, device=w.device)
            self._ver_f = None
        if with_dgrad and (self.wp_d is None or self.wp_d.dtype != dt):
            self.wp_d = torch.empty(size(Cout, Cin, ks, 1), dtype=dt, device=w.device)
            self._ver_d = None

    def out_hw(self, H, W):
        return ops.conv_out_hw(H, W, self.ks, self.stride)

    def fwd(self, x: TS, y: TS, relu=False, res: Optional[TS] = None):
        ops.conv2d(x, self.packed(0), y, self.ks, self.stride, bias=self.bsrc(), res=res, relu=relu)

    def fwd_stats(self, ctx: Ctx, x: TS, y: TS):
        """conv whose epilogue also emits the per-wave partial column sums / sums of squares of y: the BatchNorm statistics of
        the following layer without a second pass over y.  Returns (psum, psumsq, rows)."""
        assert self.conv.bias is None
        wp = self.packed(0)
        rows = ops.conv_colsum_rows(x, wp, y, self.ks, self.stride, 0)
        part = ctx.workspace(2 * rows * y.C)
        cs, cq = part[:rows * y.C], part[rows * y.C:]
        ops.conv2d(x, wp, y, self.ks, self.stride, colsum=cs, colsumsq=cq)
        return cs, cq, rows

    def bwd_w(self, ctx: Ctx, x: TS, dy: TS):
        """weight (+bias) gradient into the .grad views of the flat gradient buffer -- on the side stream when ctx.wgrad_overlap (the
        caller goes on with the input gradient; ctx.side_join() before anything reads the .grad views)"""
        side = ctx.side()
        if side is None or not ctx.overlap_now():
            self._bwd_w(ctx, x, dy, ctx.workspace)
            return
        main = ctx.main_stream or torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)                      # dy is complete on the main stream (x is a forward activation: long complete)
        side.wait_event(ready)
        if self.gapped or ctx._side_ws is None:         # (torch ops inside: they need torch's own notion of the current stream)
            with torch.cuda.stream(side):
                self._bwd_w(ctx, x, dy, ctx.side_workspace)
        else:
            ops.set_stream_override(side.cuda_stream)
            try:
                self._bwd_w(ctx, x, dy, ctx.side_workspace)
            finally:
                ops.set_stream_override(None)
        done = torch.cuda.Event()
        done.record(side)
        ctx._side_dirty = True
        ctx.side_reads(dy, done)
        ctx.side_reads(x, done)

    def _bwd_w(self, ctx: Ctx, x: TS, dy: TS, workspace):
        w, b = self.conv.weight, self.conv.bias
        key = (x.N, x.H, x.W, x.C, x.cs, dy.C, dy.cs, x.bf16, ops.wgrad_tuning_key())    # the plan (and its workspace) follows the switches' VALUES
        n = self._ws_sizes.get(key)                 # (a planning call per launch otherwise: the geometry decides)
        if n is None:
            n = self._ws_sizes[key] = ops.wgrad_workspace(x, dy, self.ks, self.stride, with_bias=b is not None)
        if self.gapped:          # gradient of the gapped filter, then its live rows / columns into the parameter's .grad
            self.wsrc()
            ops.conv2d_wgrad(x, dy, self._gwpad, self.ks, self.stride, workspace(n), dbias=None if b is None else self._gbpad)
            si, _ = self._segs(w.shape[1], self.in_gap)
            so, _ = self._segs(w.shape[0], self.out_gap)
            for lo, po in so:
                for li, pi in si:
                    w.grad[lo, li].copy_(self._gwpad[po, pi])
                if b is not None:
                    b.grad[lo].copy_(self._gbpad[po])
            return
        ops.conv2d_wgrad(x, dy, w.grad, self.ks, self.stride, workspace(n), dbias=None if b is None else b.grad)

    def bwd_x(self, dy: TS, dx: TS, res: Optional[TS] = None, mask: Optional[TS] = None):
        ops.conv2d_dgrad(dy, self.packed(1), dx, self.ks, self.stride, res=res, mask=mask)


class _BNExec:
    """Train-mode statistics / eval-mode coefficients / backward of one nn.BatchNorm2d holder."""

    def __init__(self, bn: nn.BatchNorm2d):
        self.bn = bn
        self.C = bn.num_features

    def coeffs(self, ctx: Ctx, x: TS, partials=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """partials = (psum, psumsq, rows) from the producing conv's epilogue (train mode), else a statistics pass over x"""
        bn, C_ = self.bn, self.C
        scale, shift = ctx.vec(self, "scale", C_), ctx.vec(self, "shift", C_)
        if ctx.training:
            P = x.P
            if partials is None:
                rows = ops.bn_stats_rows(P)
                part = ctx.workspace(2 * rows * C_)
                ops.bn_stats(x, part)
                ps, pq = part, part[rows * C_:]
            else:
                ps, pq, rows = partials
            mean, invstd = ctx.vec(self, "mean", C_), ctx.vec(self, "invstd", C_)
            ops.bn_finalize(ps, pq, rows, P, C_, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var,
                            BN_MOM, BN_EPS, scale, shift, mean, invstd, bn.num_batches_tracked)
            ctx.bn_epoch += 1               # running statistics moved: eval coefficients / folded filters are stale
            self._eval_key = None
        else:
            # eval coefficients depend on the parameters and the running statistics only: one launch per change, not per forward
            key = (ctx.weights_epoch, ctx.bn_epoch, bn.weight.data_ptr(), bn.running_mean.data_ptr(), bn.weight._version, bn.bias._version,
                   bn.running_mean._version, bn.running_var._version)
            if getattr(self, "_eval_key", None) != key:
                ops.bn_eval_coeffs(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, BN_EPS, scale, shift)
                self._eval_key = key
        return scale, shift

    def bwd(self, ctx: Ctx, dout: TS, out: Optional[TS], x: TS, dx: TS, gout: Optional[TS] = None, g_accumulate=False):
        """dx = dL/dx of y = bn(x) given dL/d(act(y + ..)) = dout; `out` (post-ReLU) supplies the ReLU mask."""
        bn, C_ = self.bn, self.C
        P = x.P
        mean, invstd = ctx.vec(self, "mean", C_), ctx.vec(self, "invstd", C_)
        rows = ops.bn_stats_rows(P)
        part = ctx.workspace(2 * rows * C_)
        ops.bn_bwd_reduce(dout, out, x, mean, invstd, part)
        c1, c2 = ctx.vec(self, "c1", C_), ctx.vec(self, "c2", C_)
        ops.bn_bwd_finalize(part, rows, P, C_, bn.weight.grad, bn.bias.grad, c1, c2)
        ops.bn_bwd_apply(dout, out, x, mean, invstd, bn.weight.data, c1, c2, dx, gout=gout, g_accumulate=g_accumulate)


# --------------------------------------------------------------------------
# fastai layers.py: ConvLayer / ResBlock / PixelShuffle_ICNR
# --------------------------------------------------------------------------

class ConvLayer(nn.Sequential):
    """conv [-> BN] [-> ReLU]; norm 'batch' | 'batchzero' (encoder) or None (decoder: bias, no norm)."""
    forward = _no_forward

    def __init__(self, ni, nf, ks=3, stride=1, norm: Optional[str] = "batch", act=True, bias_std=0.01, xtra: Optional[nn.Module] = None):
        bn = norm in ("batch", "batchzero")
        conv = nn.Conv2d(ni, nf, ks, stride=stride, padding=(ks - 1) // 2, bias=not bn)
        with torch.no_grad():
            if conv.bias is not None:
                conv.bias.normal_(0, bias_std) if bias_std != 0 else conv.bias.zero_()
            if act:
                nn.init.kaiming_uniform_(conv.weight)
        layers: List[nn.Module] = [conv]
        if bn:
            b = nn.BatchNorm2d(nf, eps=BN_EPS, momentum=BN_MOM)
            with torch.no_grad():
                b.bias.fill_(1e-3)
                b.weight.fill_(0.0 if norm == "batchzero" else 1.0)
            layers.append(b)
        if act:
            layers.append(nn.ReLU())
        if xtra is not None:
            layers.append(xtra)
        super().__init__(*layers)
        self.has_bn, self.has_act = bn, act
        self.cx = _ConvExec(conv)
        self.bx = _BNExec(self[1]) if bn else None
        self.cx.fold = self.bx
        self.nf = nf

    # ---- encoder flavour: conv -> BN -> [ReLU]; returns the materialised activation
    def hip_fwd(self, ctx: Ctx, x: TS) -> TS:
        OH, OW = self.cx.out_hw(x.H, x.W)
        if self.has_bn and ctx.fold_bn:      # eval: conv(x, w * scale) + shift [-> ReLU] in ONE launch
            a = ctx.act(self, "a", x.N, OH, OW, self.nf)
            self.fold_fwd(ctx, x, a)
        elif self.has_bn:
            y = ctx.act(self, "y", x.N, OH, OW, self.nf)
            scale, shift = self._conv_bn(ctx, x, y)
            a = ctx.act(self, "a", x.N, OH, OW, self.nf)
            ops.affine_act(y, a, scale, shift, relu=self.has_act)
        else:
            a = ctx.act(self, "a", x.N, OH, OW, self.nf)
            self.cx.fwd(x, a, relu=self.has_act)
        ctx.saved[(id(self), "x")] = x
        return a

    def fold_fwd(self, ctx: Ctx, x: TS, out: TS, res: Optional[TS] = None, relu: Optional[bool] = None):
        """eval mode: out = act(conv(x, w * scale) + shift [+ res]); the filter image carries the BatchNorm scale (model._pack_all)"""
        _, shift = self.bx.coeffs(ctx, None)
        ops.conv2d(x, self.cx.packed(0), out, self.cx.ks, self.cx.stride, bias=shift, res=res, relu=self.has_act if relu is None else relu)

    def raw_fwd(self, ctx: Ctx, x: TS) -> Tuple[TS, torch.Tensor, torch.Tensor]:
        """conv + BN statistics only (the affine is applied by the caller, fused with the residual add)."""
        OH, OW = self.cx.out_hw(x.H, x.W)
        y = ctx.act(self, "y", x.N, OH, OW, self.nf)
        scale, shift = self._conv_bn(ctx, x, y)
        ctx.saved[(id(self), "x")] = x
        return y, scale, shift

    def _conv_bn(self, ctx: Ctx, x: TS, y: TS):
        """conv + BatchNorm coefficients; in train mode the batch statistics come out of the conv epilogue"""
        if ctx.training and FUSE_BN_STATS and ctx.act_dtype == torch.float32 and not (self.nf > 128 and 0 < self.nf % 128 <= 64):    # (not for split launches)
            return self.bx.coeffs(ctx, y, self.cx.fwd_stats(ctx, x, y))
        self.cx.fwd(x, y)
        return self.bx.coeffs(ctx, y)

    def hip_bwd(self, ctx: Ctx, da: TS, need_dx=True, dx_res: Optional[TS] = None) -> Optional[TS]:
        """BN flavour: da = dL/d(output).  Returns dL/dx (+ dx_res fused) or None."""
        assert self.has_bn
        x: TS = ctx.saved[(id(self), "x")]
        y = ctx.act(self, "y", da.N, da.H, da.W, self.nf)
        a = ctx.act(self, "a", da.N, da.H, da.W, self.nf)
        dy = ctx.tmp(da.N, da.H, da.W, self.nf)
        self.bx.bwd(ctx, da, a if self.has_act else None, y, dy)
        dx = self.bwd_from_dy(ctx, dy, need_dx, dx_res)
        ctx.free(dy)
        return dx

    def bwd_from_dy(self, ctx: Ctx, dy: TS, need_dx=True, dx_res: Optional[TS] = None, mask: Optional[TS] = None,
                    dx_channels: Optional[int] = None) -> Optional[TS]:
        """dy = dL/d(conv output).  wgrad (+bias) then dgrad; `mask` fuses the ReLU backward of the producer of x.
        dx_channels: only the first dx_channels input channels need a gradient (the rest is the network input).
        The returned dx is a ctx.tmp() buffer owned by the caller (ctx.free it after its last reader); dy stays the caller's."""
        x: TS = ctx.saved[(id(self), "x")]
        self.cx.bwd_w(ctx, x, dy)
        if not need_dx:
            return None
        dx = ctx.tmp(x.N, x.H, x.W, x.C)
        if dx_channels is not None and dx_channels < x.C and (dx_channels + 127) // 128 == (x.C + 127) // 128:
            # same packed-filter padding: the kernel simply produces fewer channels
            self.cx.bwd_x(dy, dx.sub(0, dx_channels), res=None if dx_res is None else dx_res.sub(0, dx_channels),
                          mask=None if mask is None else mask.sub(0, dx_channels))
        else:
            self.cx.bwd_x(dy, dx, res=dx_res, mask=mask)
        return dx


class ResBlock(nn.Module):
    """fastai ResBlock.  Encoder flavour (norm='batch'): out = relu(bn(convpath(x)) + idpath(x));
    decoder flavour (norm=None, the final block): out = relu(conv2(relu(conv1(x)+b1)) + b2 + x)."""
    forward = _no_forward

    def __init__(self, expansion, ni, nf, stride=1, norm: Optional[str] = "batch"):
        super().__init__()
        norm2 = "batchzero" if norm == "batch" else norm
        nh = nf
        nf, ni = nf * expansion, ni * expansion
        if expansion == 1:
            convpath = [ConvLayer(ni, nh, 3, stride=stride, norm=norm), ConvLayer(nh, nf, 3, norm=norm2, act=False)]
        else:
            convpath = [ConvLayer(ni, nh, 1, norm=norm), ConvLayer(nh, nh, 3, stride=stride, norm=norm),
                        ConvLayer(nh, nf, 1, norm=norm2, act=False)]
        self.convpath = nn.Sequential(*convpath)
        idpath: List[nn.Module] = []
        if ni != nf:
            idpath.append(ConvLayer(ni, nf, 1, norm="batch", act=False))
        if stride != 1:
            idpath.insert(0, nn.AvgPool2d(stride, ceil_mode=True))
        self.idpath = nn.Sequential(*idpath)
        self.act = nn.ReLU(inplace=True)
        self.has_norm = norm is not None
        self.pool = stride != 1
        # plain attribute (not a registered child: it already lives in self.idpath)
        self.__dict__["idconv"] = next((m for m in self.idpath if isinstance(m, ConvLayer)), None)
        self.nf = nf

    # ------------------------------------------------------------ encoder flavour
    def hip_fwd(self, ctx: Ctx, x: TS) -> TS:
        if not self.has_norm:
            return self._fwd_nonorm(ctx, x)
        h = x
        for cl in list(self.convpath)[:-1]:
            h = cl.hip_fwd(ctx, h)
        p = x
        if self.pool:
            p = ctx.act(self, "pool", x.N, (x.H + 1) // 2, (x.W + 1) // 2, x.C)
            ops.avgpool(x, p)
        if ctx.fold_bn:         # eval: both BatchNorms folded; out = relu(conv_last(h) + shift + idpath(x)) is the last conv's epilogue
            last: ConvLayer = self.convpath[-1]
            res = p
            if self.idconv is not None:
                res = ctx.act(self.idconv, "a", p.N, p.H, p.W, self.nf)
                self.idconv.fold_fwd(ctx, p, res, relu=False)
            out = ctx.act(self, "out", res.N, res.H, res.W, self.nf)
            last.fold_fwd(ctx, h, out, res=res, relu=True)
            return out
        y2, s2, b2 = self.convpath[-1].raw_fwd(ctx, h)
        out = ctx.act(self, "out", y2.N, y2.H, y2.W, self.nf)
        if self.idconv is not None:
            yi, si, bi = self.idconv.raw_fwd(ctx, p)
            ops.affine_act(y2, out, s2, b2, x2=yi, scale2=si, shift2=bi, relu=True)
        else:
            ops.affine_act(y2, out, s2, b2, x2=p, relu=True)
        ctx.saved[(id(self), "x")] = x
        return out

    def hip_bwd(self, ctx: Ctx, dout: TS, need_dx=True) -> Optional[TS]:
        if not self.has_norm:
            raise RuntimeError("use bwd_nonorm for the decoder ResBlock")
        x: TS = ctx.saved[(id(self), "x")]
        last: ConvLayer = self.convpath[-1]
        out = ctx.act(self, "out", dout.N, dout.H, dout.W, self.nf)
        y2 = ctx.act(last, "y", dout.N, dout.H, dout.W, self.nf)
        dy2 = ctx.tmp(dout.N, dout.H, dout.W, self.nf)
        # identity branch first so that its gradient can be fused into the conv path's last dgrad
        if self.idconv is not None:
            last.bx.bwd(ctx, dout, out, y2, dy2)
            yi = ctx.act(self.idconv, "y", dout.N, dout.H, dout.W, self.nf)
            dyi = ctx.tmp(dout.N, dout.H, dout.W, self.nf)
            self.idconv.bx.bwd(ctx, dout, out, yi, dyi)
            dp = self.idconv.bwd_from_dy(ctx, dyi, need_dx=need_dx)
            ctx.free(dyi)
        else:
            dp = ctx.tmp(dout.N, dout.H, dout.W, self.nf)
            last.bx.bwd(ctx, dout, out, y2, dy2, gout=dp)
        dxb = dp
        if need_dx and self.pool:
            dxb = ctx.tmp(x.N, x.H, x.W, x.C)
            ops.avgpool_bwd(dp, dxb)
            ctx.free(dp)
        # conv path, last to first
        d = last.bwd_from_dy(ctx, dy2)
        ctx.free(dy2)
        cls = list(self.convpath)[:-1]
        for i in range(len(cls) - 1, -1, -1):
            first = i == 0
            dn = cls[i].hip_bwd(ctx, d, need_dx=(need_dx or not first), dx_res=dxb if first else None)
            ctx.free(d)
            d = dn
        ctx.free(dxb)                  # consumed as the residual of the first conv's input gradient (or unused when not need_dx)
        return d

    # ------------------------------------------------------------ decoder flavour (no norm)
    def _fwd_nonorm(self, ctx: Ctx, x: TS) -> TS:
        c1, c2 = self.convpath[0], self.convpath[1]
        t1 = ctx.act(c1, "a", x.N, x.H, x.W, c1.nf)
        c1.cx.fwd(x, t1, relu=True)
        out = ctx.act(self, "out", x.N, x.H, x.W, self.nf)
        c2.cx.fwd(t1, out, relu=True, res=x)
        ctx.saved[(id(c1), "x")] = x
        ctx.saved[(id(c2), "x")] = t1
        return out

    def bwd_nonorm(self, ctx: Ctx, dout_pre: TS, dx_channels: Optional[int] = None) -> TS:
        """dout_pre = dL/d(pre-activation of the block output), i.e. already masked by (out > 0).
        Returns dL/dx (x is not a ReLU output here: it is the dense concat); only its first dx_channels channels are
        computed when given (the trailing channels of the concat are the network input)."""
        c1, c2 = self.convpath[0], self.convpath[1]
        t1: TS = ctx.saved[(id(c2), "x")]
        dt1 = c2.bwd_from_dy(ctx, dout_pre, mask=t1)
        dx = c1.bwd_from_dy(ctx, dt1, dx_res=dout_pre, dx_channels=dx_channels)
        ctx.free(dt1)
        return dx


class PixelShuffle_ICNR(nn.Sequential):
    """1x1 ConvLayer(ni -> 4 nf, bias, ReLU) -> PixelShuffle(2) [-> ReplicationPad2d((1,0,1,0)) -> AvgPool2d(2,1)]."""
    forward = _no_forward

    def __init__(self, ni, nf=None, blur=False):
        nf = ni if nf is None else nf
        layers: List[nn.Module] = [ConvLayer(ni, nf * 4, ks=1, norm=None, bias_std=0), nn.PixelShuffle(2)]
        with torch.no_grad():
            layers[0][0].weight.copy_(icnr_init(layers[0][0].weight.data))
        if blur:
            layers += [nn.ReplicationPad2d((1, 0, 1, 0)), nn.AvgPool2d(2, stride=1)]
        super().__init__(*layers)
        self.blur, self.nf = blur, nf

    def hip_fwd(self, ctx: Ctx, up_in: TS, dst: TS, out_hw: Tuple[int, int], tail: Optional[Tuple[TS, int]] = None) -> bool:
        """writes [blur](shuffle(relu(conv1x1(up_in)))) into `dst` (a channel slice of the concat buffer),
        nearest-resized to out_hw when the skip / input size differs (non-/32 tiles).
        tail = (slice, channel): the other half of the concat, to be copied behind dst's channels at `channel` of dst's buffer; returns True when
        this call wrote it (the fused launch appends it to its own stores), False when the caller still has to."""
        cl: ConvLayer = self[0]
        ctx.saved[(id(cl), "x")] = up_in
        # no blur (the final upsample in front of the dense merge) and no resize: conv + bias + ReLU + PixelShuffle is ONE launch that stores
        # the shuffled activation straight into the concat slice -- the un-shuffled conv output (at 16 x 512^2: 1.6 GB fp32) is never
        # written, read back or kept; the backward takes its ReLU mask from the slice itself (ops.shuffle_bwd_xmask)
        fused = (not self.blur and (2 * up_in.H, 2 * up_in.W) == tuple(out_hw) and not cl.cx.gapped and self.nf % 16 == 0
                 and FUSE_SHUFFLE and ops.conv1x1_shuffle_applies(up_in, dst))
        ctx.saved[(id(self), "fused")] = dst if fused else None
        if fused:
            # (bf16 storage only: at 16 x 256^2 x 96 -> 4 x 96 the launch + the second pass are 490 + 158 us against 598 us in one launch; in
            #  fp32 1021 + 155 against 1229 -- the appended 16-byte stores cost the launch more than the pass they replace: scripts/ab_conv_head.py)
            with_tail = (tail is not None and FUSE_CONCAT_TAIL and dst.bf16 and ops.conv1x1_shuffle_tail_ok(dst, tail[0], tail[1]))
            ops.conv1x1_shuffle(up_in, cl.cx.packed(2), dst, bias=cl.cx.bsrc(), relu=True, tail=tail[0] if with_tail else None,
                                tail_at=tail[1] if with_tail else 0)
            return with_tail
        yc = ctx.act(cl, "a", up_in.N, up_in.H, up_in.W, 4 * self.nf)
        cl.cx.fwd(up_in, yc, relu=True)
        if (2 * up_in.H, 2 * up_in.W) == tuple(out_hw):
            ops.shuffle_blur(yc, dst, self.blur)
        else:
            tmp = ctx.act(self, "up", up_in.N, 2 * up_in.H, 2 * up_in.W, self.nf)
            ops.shuffle_blur(yc, tmp, self.blur)
            ops.resize_nearest(tmp, dst)
        return False

    def hip_bwd(self, ctx: Ctx, d_dst: TS, mask_input: bool = True) -> TS:
        """d_dst = dL/d(dst slice).  Returns dL/d(pre-activation of the producer of up_in) (masked by up_in > 0 when up_in is
        a ReLU output, i.e. mask_input)."""
        cl: ConvLayer = self[0]
        up_in: TS = ctx.saved[(id(cl), "x")]
        dyc = ctx.tmp(up_in.N, up_in.H, up_in.W, 4 * self.nf)
        fused_dst: Optional[TS] = ctx.saved.get((id(self), "fused"))
        if fused_dst is not None:
            ops.shuffle_bwd_xmask(d_dst, fused_dst, dyc)
            dx = cl.bwd_from_dy(ctx, dyc, mask=up_in if mask_input else None)
            ctx.free(dyc)
            return dx
        yc = ctx.act(cl, "a", up_in.N, up_in.H, up_in.W, 4 * self.nf)
        if (2 * up_in.H, 2 * up_in.W) == (d_dst.H, d_dst.W):
            ops.shuffle_blur_bwd(d_dst, yc, dyc, self.blur)
        else:
            tmp = ctx.tmp(up_in.N, 2 * up_in.H, 2 * up_in.W, self.nf)
            ops.resize_nearest_bwd(d_dst, tmp)
            ops.shuffle_blur_bwd(tmp, yc, dyc, self.blur)
            ctx.free(tmp)
        dx = cl.bwd_from_dy(ctx, dyc, mask=up_in if mask_input else None)
        ctx.free(dyc)
        return dx


def icnr_init(x: torch.Tensor, scale=2, init=nn.init.kaiming_normal_) -> torch.Tensor:
    ni, nf, h, w = x.shape
    ni2 = int(ni / (scale ** 2))
    k = init(x.new_zeros([ni2, nf, h, w])).transpose(0, 1)
    k = k.contiguous().view(ni2, nf, -1).repeat(1, 1, scale ** 2)
    return k.contiguous().view([nf, ni, h, w]).transpose(0, 1)



# --------------------------------------------------------------------------
# fastai layers.py: SelfAttention (DynamicUnet(self_attention=True): params_and_main.py:81-83, train.py:141-144)
# --------------------------------------------------------------------------

class SelfAttention(nn.Module):
    """x -> gamma * (h beta) + x with f,g,h = spectral-normed 1x1 projections, beta = softmax(f^T g, dim=1).

    Device program (NHWC, rows = positions): QKV = X [Wq;Wk;Wv]^T in ONE 1x1 conv; the N x N products run on the MFMA conv / wgrad
    kernels with operands packed from activations (unet_pack_weights_strided):
        T[j][i] = sum_c G[j][c] F[i][c]      (= S^T)          P = row-softmax(T)  (= beta^T)
        O[j][c] = sum_i P[j][i] H[i][c]                        out = gamma * O + X
    BLOCKWISE over the rows j: a chunk is either a group of whole images (one launch per product over the group,
    unet_conv_desc.wp_img_stride) or, when one N x N matrix exceeds ``budget_elems``, a block of rows of one image.  Every row of T is
    complete inside its chunk, so the softmax is exact without online rescaling; the chunk's T / P / dP live in two scratch buffers
    of at most ``budget_elems`` floats each, and when there is more than one chunk P is NOT kept for the backward pass but recomputed
    from QKV (one extra N^2 C/8 product: the flash-attention trade).  xresnet50 at 1024 x 1024 (N = 16384, 1 GiB per N x N fp32
    matrix) therefore needs 2 x budget instead of 3 GiB per tile.
    Spectral normalisation (legacy torch.nn.utils.spectral_norm: one power iteration per training forward) acts on three
    tiny matrices and stays in torch; its backward is torch autograd on those matrices, fed with the HIP weight gradient."""
    forward = _no_forward
    budget_elems = 1 << 28          # floats per N x N scratch buffer (1 GiB); tests lower it to force row blocks
    fused = True                    # bf16 storage, C = 384: the fused kernels of csrc/attention.hip (no N x N tensor); False = blockwise products; "always" = every width the library takes

    def __init__(self, n_channels):
        super().__init__()
        c8 = n_channels // 8
        self.query = nn.Sequential(nn.utils.spectral_norm(nn.Conv1d(n_channels, c8, 1, bias=False)))
        self.key = nn.Sequential(nn.utils.spectral_norm(nn.Conv1d(n_channels, c8, 1, bias=False)))
        self.value = nn.Sequential(nn.utils.spectral_norm(nn.Conv1d(n_channels, n_channels, 1, bias=False)))
        self.gamma = nn.Parameter(torch.tensor([0.0]))
        self.C, self.c8 = n_channels, c8
        # the query / key / value slices of the fused QKV buffer are channel slices and start at multiples of the 4-channel vector:
        # query and key are padded to c8p lanes (xresnet34_deep: 432 channels, 54 -> 56) with zero filter rows, i.e. zero lanes that
        # add nothing to any product
        self.c8p = (c8 + 3) // 4 * 4          # fp32 storage (4-lane vectors); bf16 storage pads to 8 lanes (ops.rupv at run time)
        if n_channels % 8 != 0:
            raise ValueError(f"self-attention on {n_channels} channels: the device program needs a multiple of 8 channels")

    @staticmethod
    def _normed_weight(seq: nn.Sequential) -> torch.Tensor:
        m = seq[0]
        for hook in m._forward_pre_hooks.values():      # SpectralNorm.__call__: power iteration (training) + W / sigma
            hook(m, None)
        return m.weight

    # ---- chunking: (b0, nb, j0, nj) = images [b0, b0+nb) x rows [j0, j0+nj) of their N x N attention matrix
    def _chunks(self, B, H, W):
        N, budget = H * W, int(self.budget_elems)
        if N * N <= budget:
            nb = max(1, min(B, budget // (N * N)))
            return [(b0, min(nb, B - b0), 0, N) for b0 in range(0, B, nb)]
        unit = 4 * W                                     # whole image rows, a multiple of the conv kernel's 4-row pixel tile
        rows = min(N, max(unit, budget // N // unit * unit))
        return [(b, 1, j0, min(rows, N - j0)) for b in range(B) for j0 in range(0, N, rows)]

    @staticmethod
    def _rows(buf: torch.Tensor, co: int, C: int, b0: int, nb: int, j0: int, nj: int, W: int) -> TS:
        """rows [j0, j0+nj) of images [b0, b0+nb) of an activation buffer as a TS over [nb, nj / W, W] pixels"""
        if j0 == 0 and nj == buf.shape[1] * buf.shape[2]:
            return TS(buf[b0:b0 + nb], co, C)
        cs = buf.shape[3]
        return TS(buf[b0].view(-1, cs)[j0:j0 + nj].view(1, nj // W, W, cs), co, C)

    def _scratch(self, ctx: Ctx, tag: str, chunks, N: int, nb: int, nj: int, W: int, dtype=torch.float32) -> TS:
        """[nb, nj / W, W, Np] view of a scratch buffer holding one N-wide row per attention row of the chunk.  Rows are Np = roundup(N, 8)
        elements long (16-byte rows in both storage types: 50 x 50 = 2500 positions of the reference's 400-px tiles are not a multiple of 8);
        the pad lanes of the view are zeroed, the kernels read them as part of the last channel vector"""
        Np = (N + 7) // 8 * 8
        big = max(c[1] * c[3] for c in chunks) * Np
        buf = ctx.vec(self, tag, big, dtype=dtype)
        v = buf[:nb * nj * Np].view(nb, nj // W, W, Np)
        if Np != N:
            v[..., N:].zero_()
        return TS(v, 0, N)

    def _scores(self, ctx: Ctx, qkv: TS, chunks, ch, W: int, wpa: torch.Tensor) -> TS:
        """row-softmax(G_chunk F^T): the rows of beta^T that belong to the chunk.  The logits are fp32 in both storage modes (bf16: the
        product writes fp32, unet_conv_desc.y_f32); the weights come back in the activation storage type (fp32: in place)."""
        b0, nb, j0, nj = ch
        N = qkv.buf.shape[1] * qkv.buf.shape[2]
        c8, dt = self.c8, qkv.buf.dtype
        c8p, CQ, es = ops.rupv(c8, dt), qkv.cs, qkv.buf.element_size()
        img = N * CQ * es
        sz = ops.pack_size(N, c8, dt)
        for k in range(nb):
            ops.pack_weights_strided(qkv.ptr + (b0 + k) * img, CQ, 1, N, c8, wpa[k * sz:])                 # (o=i, r=c) = F_b[i][c]
        T = self._scratch(ctx, "TP", chunks, N, nb, nj, W)
        ops.conv2d(self._rows(qkv.buf, c8p, c8, b0, nb, j0, nj, W), wpa, T, 1, wp_img_stride=sz)          # T = G F^T
        if dt == torch.float32:
            ops.row_softmax(T, T)                                                                          # in place
            return T
        P = self._scratch(ctx, "P16", chunks, N, nb, nj, W, dtype=dt)
        ops.row_softmax(T, P)
        return P

    def _wcat(self, c8p: int) -> torch.Tensor:
        wq_, wk_, wv_ = self._normed_weight(self.query), self._normed_weight(self.key), self._normed_weight(self.value)
        if c8p != self.c8:
            zpad = wq_.new_zeros((c8p - self.c8,) + tuple(wq_.shape[1:]))
            return torch.cat([wq_, zpad, wk_, zpad, wv_], 0)
        return torch.cat([wq_, wk_, wv_], 0)

    def hip_fwd(self, ctx: Ctx, x: TS) -> TS:
        B, H, W, C_, c8, dt = x.N, x.H, x.W, self.C, self.c8, ctx.act_dtype
        c8p, es = ops.rupv(c8, dt), x.buf.element_size()      # query / key slices padded to the vector width (fp32: 4 lanes, bf16: 8)
        N, CQ = H * W, 2 * c8p + C_
        with torch.enable_grad():
            wcat = self._wcat(c8p).reshape(CQ, C_, 1, 1)
        ctx.saved[(id(self), "wcat")] = wcat
        wq = wcat.detach().contiguous()
        wp = ctx.vec(self, "wp_f", ops.pack_size(CQ, C_, dt), dtype=dt)
        ops.pack_weights(wq, 0, wp, dtype=dt)
        qkv = ctx.act(self, "qkv", B, H, W, CQ)
        ops.conv2d(x, wp, qkv, 1)
        O = ctx.act(self, "O", B, H, W, C_)
        # the fused kernels where they are faster: the instantiations without per-tile guards (C = 384, the width of every xresnet18 / 34 network).
        # Other widths the library accepts (C <= 512: xresnet34_deep's 432) run its guarded instantiations 5 % SLOWER than the blockwise products
        # (361 vs 378 tiles/s at cfg2 geometry, batch 8): they stay blockwise unless fused == "always" (the kernel tests call the library directly)
        use_fused = bool(self.fused) and dt == torch.bfloat16 and ops.sa_fused_supported(c8p, C_) and ((C_ == 384 and c8p <= 56) or self.fused == "always")
        ctx.saved[(id(self), "fused")] = use_fused
        chunks = [] if use_fused else self._chunks(B, H, W)
        if use_fused:
            # QK^T -> softmax -> . V in one launch (csrc/attention.hip); lse kept for the backward pass, which recomputes the weights
            pk = ctx.vec(self, "sa_pack", B * ops.sa_pack_elems(N, C_), dtype=dt)
            ops.sa_pack(qkv.sub(2 * c8p, C_), pk)
            ops.sa_fwd(qkv, c8p, C_, pk, O, ctx.vec(self, "sa_lse", B * ops.sa_rows(N)))
        else:
            nbmax = max(c[1] for c in chunks)
            szs = [ops.pack_size(N, c8, dt), ops.pack_size(C_, N, dt), ops.pack_size(N, C_, dt), ops.pack_size(c8p, N, dt)]
            wpa = ctx.vec(self, "wp_a", nbmax * max(szs), dtype=dt)
        img = N * CQ * es     # bytes per image of qkv
        for ch in chunks:
            b0, nb, j0, nj = ch
            P = self._scores(ctx, qkv, chunks, ch, W, wpa)
            for k in range(nb):
                ops.pack_weights_strided(qkv.ptr + (b0 + k) * img + 2 * c8p * es, 1, CQ, C_, N, wpa[k * szs[1]:])   # (o=c, r=i) = H_b[i][c]
            ops.conv2d(P, wpa, self._rows(O.buf, 0, C_, b0, nb, j0, nj, W), 1, wp_img_stride=szs[1])                 # O = P H
        gvec, zvec = ctx.vec(self, "gvec", C_), ctx.vec(self, "zvec", C_)
        gvec.copy_(self.gamma.data.expand(C_))
        out = ctx.act(self, "out", B, H, W, C_)
        ops.affine_act(O, out, gvec, zvec, x2=x)
        ctx.saved[(id(self), "x")] = x
        ctx.saved[(id(self), "P_kept")] = len(chunks) == 1 and not use_fused      # a single chunk leaves its P in the scratch buffer for the backward
        return out

    def hip_bwd(self, ctx: Ctx, dout: TS) -> TS:
        """dout = dL/d(out).  Returns dL/dx."""
        x: TS = ctx.saved[(id(self), "x")]
        B, H, W, C_, c8, dt = x.N, x.H, x.W, self.C, self.c8, ctx.act_dtype
        c8p, es, bf = ops.rupv(c8, dt), x.buf.element_size(), ctx.act_dtype == torch.bfloat16
        N, CQ = H * W, 2 * c8p + C_
        qkv, O = ctx.act(self, "qkv", B, H, W, CQ), ctx.act(self, "O", B, H, W, C_)
        gvec, zvec = ctx.vec(self, "gvec", C_), ctx.vec(self, "zvec", C_)
        use_fused = bool(ctx.saved.get((id(self), "fused"), False))
        if use_fused:
            # one pass over dout and O gives both the per-row term of the softmax adjoint and dL/dgamma: Dp_j = dout_j . O_j,
            # dgamma = sum_j Dp_j, D_j = dO_j . O_j = gamma Dp_j (two torch launches on B x N floats instead of a 50 MB dot + its reduction)
            lse, D = ctx.vec(self, "sa_lse", B * ops.sa_rows(N)), ctx.vec(self, "sa_D", B * ops.sa_rows(N))
            ops.sa_rowdot(dout, O, D)                                                         # (rows past N stay 0)
            torch.sum(D.view(1, -1), dim=1, out=self.gamma.grad)
            D.mul_(self.gamma.data)
        else:
            ops.dot(O, dout, self.gamma.grad, ctx.workspace(ops.colsum_workspace(O.P, C_)))    # dL/dgamma = sum O * dout
        dO = ctx.tmp(B, H, W, C_)
        ops.affine_act(dout, dO, gvec, zvec)
        dqkv = ctx.tmp(B, H, W, CQ)
        chunks = [] if use_fused else self._chunks(B, H, W)
        if use_fused:
            # dH, dF per key block and dG per query block in two launches, weights recomputed from lse; the packed images are the
            # operands of the products that sum over positions (dO for dH, G for dF, F for dG)
            pk = ctx.vec(self, "sa_pack", B * ops.sa_pack_elems(N, C_), dtype=dt)               # the forward's image of H is not needed any more
            gpk, fpk = (ctx.vec(self, t, B * ops.sa_pack_elems(N, c8p), dtype=dt) for t in ("sa_gpack", "sa_fpack"))
            ops.sa_pack(dO, pk)
            ops.sa_pack(qkv.sub(c8p, c8p), gpk)
            ops.sa_pack(qkv.sub(0, c8p), fpk)
            ops.sa_bwd(qkv, c8p, C_, dO, pk, gpk, fpk, lse, D, dqkv)
        else:
            kept = bool(ctx.saved.get((id(self), "P_kept"), False)) and len(chunks) == 1
            nbmax = max(c[1] for c in chunks)
            szs = [ops.pack_size(N, c8, dt), ops.pack_size(C_, N, dt), ops.pack_size(N, C_, dt), ops.pack_size(c8p, N, dt)]
            wpa = ctx.vec(self, "wp_a", nbmax * max(szs), dtype=dt)
            tmpH, tmpF = ctx.vec(self, "tmpH", N * C_), ctx.vec(self, "tmpF", N * c8p)          # weight-gradient results: fp32 in both modes
            put = ops.cast_slice if bf else ops.copy_slice                                      # ... into the gradient of the QKV tensor
        img = N * CQ * es
        for ch in chunks:
            b0, nb, j0, nj = ch
            if kept:
                P = self._scratch(ctx, "P16" if bf else "TP", chunks, N, nb, nj, W, dtype=dt)
            else:
                P = self._scores(ctx, qkv, chunks, ch, W, wpa)                                   # recompute beta^T rows
            dP = self._scratch(ctx, "dP", chunks, N, nb, nj, W)                                  # fp32 (bf16: written by the product as fp32)
            dO_c = self._rows(dO.buf, 0, C_, b0, nb, j0, nj, W)
            for k in range(nb):
                ops.pack_weights_strided(qkv.ptr + (b0 + k) * img + 2 * c8p * es, CQ, 1, N, C_, wpa[k * szs[2]:])  # (o=i, r=c) = H_b[i][c]
            ops.conv2d(dO_c, wpa, dP, 1, wp_img_stride=szs[2])                                                     # dP = dO H^T
            last = j0 + nj == N
            for k in range(nb):
                b = b0 + k
                dO_b, P_b = self._rows(dO.buf, 0, C_, b, 1, j0, nj, W), TS(P.buf[k:k + 1], 0, N)
                n = ops.wgrad_workspace(dO_b, P_b, 1, 1)
                ops.conv2d_wgrad(dO_b, P_b, tmpH, 1, 1, ctx.workspace(n), accumulate=j0 > 0)                       # dH_b (+)= P^T dO -> [N][C]
                if last:
                    put(TS(tmpH[:N * C_].view(1, H, W, C_), 0, C_), TS(dqkv.buf[b:b + 1], 2 * c8p, C_))
            if bf:
                dT = self._scratch(ctx, "dT16", chunks, N, nb, nj, W, dtype=dt)
                ops.row_softmax_bwd(P, dP, dT)
            else:
                dT = dP
                ops.row_softmax_bwd(P, dP, dP)                                                                     # dT in place
            for k in range(nb):
                ops.pack_weights_strided(qkv.ptr + (b0 + k) * img, 1, CQ, c8p, N, wpa[k * szs[3]:])                # (o=c, r=i) = F_b[i][c]
            # c8p output channels: the pad lanes of F are zeros, so the pad lanes of dG are written as exact zeros (dqkv is pool memory)
            ops.conv2d(dT, wpa, self._rows(dqkv.buf, c8p, c8p, b0, nb, j0, nj, W), 1, wp_img_stride=szs[3])        # dG = dT F
            for k in range(nb):
                b = b0 + k
                # G with its zero pad lanes (c8p wide): the gradient rows come out 16-byte aligned, their pad lanes are exact zeros
                dT_b, G_b = TS(dT.buf[k:k + 1], 0, N), self._rows(qkv.buf, c8p, c8p, b, 1, j0, nj, W)
                n = ops.wgrad_workspace(G_b, dT_b, 1, 1)
                ops.conv2d_wgrad(G_b, dT_b, tmpF, 1, 1, ctx.workspace(n), accumulate=j0 > 0)                       # dF_b (+)= dT^T G -> [N][c8p]
                if last:
                    put(TS(tmpF[:N * c8p].view(1, H, W, c8p), 0, c8p), TS(dqkv.buf[b:b + 1], 0, c8p))
        # back through the fused QKV projection
        wcat: torch.Tensor = ctx.saved[(id(self), "wcat")]
        dw = ctx.vec(self, "dwcat", CQ * C_).view(CQ, C_, 1, 1)
        n = ops.wgrad_workspace(x, dqkv, 1, 1)
        ops.conv2d_wgrad(x, dqkv, dw, 1, 1, ctx.workspace(n))
        wpd = ctx.vec(self, "wp_d", int((ops.lib.unet_pack_weights_size_bf16 if bf else ops.lib.unet_pack_weights_size)(CQ, C_, 1, 1)), dtype=dt)
        ops.pack_weights(wcat.detach().contiguous(), 1, wpd, dtype=dt)
        dx = ctx.tmp(B, H, W, C_)
        ops.conv2d_dgrad(dqkv, wpd, dx, 1, 1, res=dout)                                           # + identity branch
        ctx.free(dO)
        ctx.free(dqkv)
        # spectral-norm backward (torch, tiny).  torch.autograd.grad, not .backward(): no AccumulateGrad nodes (they remember the stream of their
        # first use and warn about it on any other stream).  A hipGraph capture of a step with this module still dies inside hipStreamEndCapture on
        # ROCm 7.2 / torch 2.10 (with .backward(), with .grad(), with the engine single-threaded; blockwise and fused, fp32 and bf16):
        # TrainStep(use_graph=True) therefore runs models with SelfAttention eagerly (trainer.py)
        leaves = [seq[0].weight_orig for seq in (self.query, self.key, self.value)]
        for p, g in zip(leaves, torch.autograd.grad(wcat, leaves, dw)):
            p.grad.copy_(g)
        return dx

# --------------------------------------------------------------------------
# decoder blocks (vision/models/unet.py)
# --------------------------------------------------------------------------

def _bias_relu_layer_fwd(ctx: Ctx, cl: ConvLayer, x: TS) -> TS:
    a = ctx.act(cl, "a", x.N, x.H, x.W, cl.nf)
    cl.cx.fwd(x, a, relu=True)
    ctx.saved[(id(cl), "x")] = x
    return a


class UnetBlock(nn.Module):
    forward = _no_forward

    def __init__(self, up_in_c, x_in_c, final_div=True, blur=True, self_attention=False, up_is_relu=True):
        super().__init__()
        self.shuf = PixelShuffle_ICNR(up_in_c, up_in_c // 2, blur=blur)
        self.bn = nn.BatchNorm2d(x_in_c, eps=BN_EPS, momentum=BN_MOM)
        with torch.no_grad():
            self.bn.bias.fill_(1e-3)
        ni = up_in_c // 2 + x_in_c
        nf = ni if final_div else ni // 2
        self.conv1 = ConvLayer(ni, nf, norm=None)
        self.conv2 = ConvLayer(nf, nf, norm=None, xtra=SelfAttention(nf) if self_attention else None)
        self.relu = nn.ReLU()
        _kaiming_init(self.conv1, self.conv2[0])
        self.cu, self.cs, self.ni, self.out_channels = up_in_c // 2, x_in_c, ni, nf
        self.cu_off, self.ni_p = self.cu, ni       # physical offset of the skip slice / width of the concat buffer (set_gap: bf16 storage)
        self.bx = _BNExec(self.bn)
        self.__dict__["sa"] = self.conv2[2] if self_attention else None
        # is the tensor this block up-samples a ReLU output (then the ReLU backward is fused as a dgrad mask)?  Not when the
        # previous block ends in self-attention.
        self.up_is_relu = up_is_relu

    def set_gap(self, vec: int):
        """concat buffer [up (cu) | gap | skip (cs)] with the skip slice at a multiple of `vec` channels; conv1 reads it with zero filters
        for the gap channels (the up slice owns them as its padding: the shuffle kernel writes zeros there)"""
        self.cu_off = (self.cu + vec - 1) // vec * vec
        self.ni_p = self.cu_off + self.cs
        self.conv1.cx.set_gaps(in_gap=(self.cu, self.cu_off - self.cu))

    def hip_fwd(self, ctx: Ctx, up_in: TS, s: TS) -> TS:
        X = ctx.act(self, "cat", s.N, s.H, s.W, self.ni_p, zero=self.ni_p != self.ni)
        self.shuf.hip_fwd(ctx, up_in, X.sub(0, self.cu), (s.H, s.W))
        scale, shift = self.bx.coeffs(ctx, s)
        ops.affine_act(s, X.sub(self.cu_off, self.cs), scale, shift, relu=True)
        ctx.saved[(id(self), "s")] = s
        t1 = _bias_relu_layer_fwd(ctx, self.conv1, X)
        t2 = _bias_relu_layer_fwd(ctx, self.conv2, t1)
        return t2 if self.sa is None else self.sa.hip_fwd(ctx, t2)

    def hip_bwd(self, ctx: Ctx, dt2_pre: TS, dskip: TS, dskip_accumulate: bool) -> TS:
        """dt2_pre = masked gradient w.r.t. conv2's pre-activation (or, when the block ends in self-attention, the plain
        gradient w.r.t. the block output).  Writes dL/d(skip) into dskip and returns the gradient for the producer of up_in
        (masked by its ReLU when up_is_relu)."""
        s: TS = ctx.saved[(id(self), "s")]
        X = ctx.act(self, "cat", s.N, s.H, s.W, self.ni_p)
        t1: TS = ctx.saved[(id(self.conv2), "x")]
        own = None
        if self.sa is not None:
            t2 = ctx.act(self.conv2, "a", s.N, s.H, s.W, self.out_channels)
            d_t2 = self.sa.hip_bwd(ctx, dt2_pre)
            dt2_pre = own = ctx.tmp(s.N, s.H, s.W, self.out_channels)
            ops.relu_mask(d_t2, t2, dt2_pre)
            ctx.free(d_t2)
        dt1 = self.conv2.bwd_from_dy(ctx, dt2_pre, mask=t1)
        ctx.free(own)
        dX = self.conv1.bwd_from_dy(ctx, dt1, mask=X)          # relu(cat) backward fused
        ctx.free(dt1)
        assert not dskip_accumulate
        self.bx.bwd(ctx, dX.sub(self.cu_off, self.cs), None, s, dskip)
        d = self.shuf.hip_bwd(ctx, dX.sub(0, self.cu), mask_input=self.up_is_relu)
        ctx.free(dX)
        return d


def _kaiming_init(*mods):
    for m in mods:
        for l in m.modules():
            if isinstance(l, nn.Conv2d):
                nn.init.kaiming_normal_(l.weight)
                if l.bias is not None:
                    with torch.no_grad():
                        l.bias.fill_(0.0)


# the constructors the reference imports (params_and_main.py:12): expansion, blocks per stage; stage widths of fastai's XResNet are
# [64, 128, 256, 512] + [256] * (len(layers) - 4), every stage after the first halves the resolution
XRESNET_LAYERS = {"xresnet18": (1, [2, 2, 2, 2]), "xresnet34": (1, [3, 4, 6, 3]), "xresnet50": (4, [3, 4, 6, 3]),
                  "xresnet101": (4, [3, 4, 23, 3]), "xresnet34_deep": (1, [3, 4, 6, 3, 1, 1])}


def _init_cnn(m: nn.Module):
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0)
    if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
        nn.init.kaiming_normal_(m.weight)
    for l in m.children():
        _init_cnn(l)


class MaxPool(nn.MaxPool2d):
    forward = _no_forward


class Stage(nn.Sequential):
    forward = _no_forward


class Encoder(nn.Sequential):
    """children of fastai XResNet up to the pooling layer (create_body: 3 stem convs, max-pool, 4 or 6 stages) with the reference's stem swap."""
    forward = _no_forward

    def __getitem__(self, idx):
        if isinstance(idx, slice):   # m[0][:3] / m[0][3:] of the fastai splitter (train.py:78-80)
            return nn.Sequential(*list(self._modules.values())[idx])
        return super().__getitem__(idx)

    def __init__(self, arch: str, c_in: int):
        expansion, layers = XRESNET_LAYERS[arch]
        stem_szs = [3, 32, 32, 64]
        stem = [ConvLayer(stem_szs[i], stem_szs[i + 1], 3, stride=2 if i == 0 else 1) for i in range(3)]
        block_szs = [64 // expansion, 64, 128, 256, 512] + [256] * (len(layers) - 4)
        stages = []
        for i, nb in enumerate(layers):
            ni, nf = block_szs[i], block_szs[i + 1]
            stages.append(Stage(*[ResBlock(expansion, ni if j == 0 else nf, nf, stride=(1 if i == 0 else 2) if j == 0 else 1)
                                  for j in range(nb)]))
        super().__init__(*stem, MaxPool(3, stride=2, padding=1), *stages)
        _init_cnn(self)
        # train.py:130-135: fresh nn.Conv2d(c_in, 32, 3, 2, 1, bias=False) with PyTorch's default init
        new_conv = nn.Conv2d(c_in, 32, kernel_size=3, stride=2, padding=1, bias=False)
        self[0][0] = new_conv
        self[0].cx = _ConvExec(new_conv)
        self[0].cx.fold = self[0].bx
        self.out_channels = block_szs[-1] * expansion
        # DynamicUnet hooks the children whose output is larger than the next child's: the last stem conv (before the max-pool) and
        # every stage but the last (each later stage starts with a stride-2 block)
        self.skip_channels = {2: 64}
        self.skip_channels.update({4 + i: block_szs[i + 1] * expansion for i in range(len(layers) - 1)})
