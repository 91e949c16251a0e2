"""HipDynamicUnet: the MI355X-native drop-in for the model object the reference builds
with ``models.unet.DynamicUnet(body, n_out, img_size, blur=True, blur_final=True,
self_attention, y_range=None, norm_type=NormType, last_cross=True, bottle=False)``
(``train.py:141-144``) on a ``create_body(xresnetNN)`` encoder whose first conv was
replaced for N input channels (``train.py:128-135``).

Surface kept: ``nn.Module``; ``model(x: f32[B,C,H,W]) -> f32[B,n_out,H,W]``
raw logits; ``train()/eval()``; ``parameters()``; ``state_dict()/load_state_dict()`` with
fastai's key scheme; indexing ``m[0]`` / ``m[0][:3]`` / ``m[0][3:]`` / ``m[1:]``.

Execution: all parameters live in ONE flat fp32 device buffer (grads in a second one),
activations in persistent NHWC buffers; forward and backward are explicit programs of
C-ABI launches (``unet_amd/modules.py``), so a whole step can be captured in a hipGraph.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .modules import (BN_EPS, BN_MOM, ConvLayer, Ctx, Encoder, PixelShuffle_ICNR, ResBlock, UnetBlock, _BNExec, _ConvExec,
                      _kaiming_init)
from .ops import TS


class _Marker(nn.Module):
    """parameter-free stand-ins that keep fastai's child numbering (ResizeToOrig, MergeLayer)."""
    def __init__(self, name):
        super().__init__()
        self._name = name

    def extra_repr(self):
        return self._name

    def forward(self, *a, **k):
        raise RuntimeError(f"{self._name} runs only inside HipDynamicUnet")


@contextlib.contextmanager
def skip_weight_init():
    """Build a model WITHOUT drawing its random initial weights: for the duration of the block the two generators every large tensor of the
    tree goes through (`nn.init.kaiming_uniform_` in `nn.Conv2d.reset_parameters`, `nn.init.kaiming_normal_` in fastai's `init_cnn` /
    `apply_init`) leave the tensor as allocated.  For callers that overwrite every parameter and buffer right away (`load_learner`: a strict
    `load_state_dict`) -- 41 M host-side normal / uniform draws are 0.3 s of a 0.7 s `save_predictions` call over 400 tiles."""
    saved = nn.init.kaiming_uniform_, nn.init.kaiming_normal_
    nn.init.kaiming_uniform_ = nn.init.kaiming_normal_ = lambda tensor, *a, **k: tensor
    try:
        yield
    finally:
        nn.init.kaiming_uniform_, nn.init.kaiming_normal_ = saved


class HipDynamicUnet(nn.Module):
    # eval-mode forward: every BatchNorm that follows a conv is folded into that conv's packed filter (w * gamma / sqrt(var + eps)) and its
    # epilogue bias (beta - mean * scale), residual add and ReLU included: north_star's "Conv + BN + ReLU fused".  False keeps the separate
    # BatchNorm-apply pass (A/B and tests); the training path always normalises with batch statistics in its own kernels.
    fold_eval_bn = os.environ.get("UNET_FOLD_BN", "1") != "0"

    def __init__(self, arch: str, n_in: int, n_out: int, img_size: Sequence[int] = (512, 512), self_attention: bool = False,
                 device="cuda", act_dtype: str = "f32"):
        """act_dtype "f32": the parity path (the reference computes in fp32).  "bf16": bf16 storage of activations, activation
        gradients and packed filters with fp32 accumulation in the matrix cores; parameters, gradients of parameters, BatchNorm
        statistics, logits (and attention logits), loss and optimizer state stay fp32 (BASELINE.json configs[1] variant; classification)."""
        super().__init__()
        if act_dtype not in ("f32", "bf16"):
            raise ValueError(f"act_dtype must be 'f32' or 'bf16', not {act_dtype!r}")
        self.act_dtype = act_dtype
        self.self_attention = bool(self_attention)
        self.arch, self.n_in, self.n_out = arch, n_in, n_out
        self.img_size = tuple(img_size)
        enc = Encoder(arch, n_in)
        # DynamicUnet.__init__: encoder children whose output size differs from the next child's are the skips
        self.sz_chg_idxs = sorted(enc.skip_channels, reverse=True)      # xresnet18/34/50/101: [6, 5, 4, 2]; xresnet34_deep: [8, 7, 6, 5, 4, 2]
        ni = enc.out_channels
        post_bn = nn.BatchNorm2d(ni, eps=BN_EPS, momentum=BN_MOM)
        with torch.no_grad():
            post_bn.bias.fill_(1e-3)
        middle = nn.Sequential(ConvLayer(ni, ni * 2, norm=None), ConvLayer(ni * 2, ni, norm=None))
        layers: List[nn.Module] = [enc, post_bn, nn.ReLU(), middle]
        xc = ni
        prev_sa = False
        for i, idx in enumerate(self.sz_chg_idxs):
            not_final = i != len(self.sz_chg_idxs) - 1
            sa = self_attention and (i == len(self.sz_chg_idxs) - 3)
            blk = UnetBlock(xc, enc.skip_channels[idx], final_div=not_final, blur=True, self_attention=sa, up_is_relu=not prev_sa)
            layers.append(blk)
            xc = blk.out_channels
            prev_sa = sa
        layers.append(PixelShuffle_ICNR(xc))
        layers.append(_Marker("ResizeToOrig"))
        layers.append(_Marker("MergeLayer(dense=True)"))
        self.up_c = xc
        xc += n_in
        layers.append(ResBlock(1, xc, xc, norm=None))
        layers.append(ConvLayer(xc, n_out, ks=1, norm=None, act=False))
        _kaiming_init(layers[3], layers[-2])
        self.layers = nn.ModuleList(layers)
        self.cat_c = xc
        # every concat inside a UnetBlock is two channel slices of one buffer: the second starts at a multiple of the vector width.  bf16
        # storage (8-channel vectors) of an encoder whose decoder widths are not multiples of 8 (xresnet34_deep: 140 in the last UnetBlock, 102
        # in front of the dense merge) leaves a GAP of dead channels there: the consumers' filters carry zero rows / columns for them
        vec = 8 if act_dtype == "bf16" else 4
        blocks = layers[4:4 + len(self.sz_chg_idxs)]
        ragged = [b.cu for b in blocks if b.cu % vec] + ([self.up_c] if act_dtype == "bf16" and self.up_c % vec else [])      # (fp32: the dense merge takes any width)
        self.up_off, self.cat_p = self.up_c, self.cat_c          # physical offset of the network-input slice / width of the dense merge
        if ragged and act_dtype == "bf16":
            for b in blocks:
                if b.cu % vec:
                    b.set_gap(vec)
            if self.up_c % vec:
                self.up_off = (self.up_c + vec - 1) // vec * vec
                self.cat_p = self.up_off + n_in
                gap = (self.up_c, self.up_off - self.up_c)
                rb, head = layers[-2], layers[-1]
                for cl in (rb.convpath[0], rb.convpath[1]):
                    cl.cx.set_gaps(in_gap=gap, out_gap=gap)
                    cl.nf = self.cat_p
                rb.nf = self.cat_p
                head.cx.set_gaps(in_gap=gap)
        elif ragged:
            raise ValueError(f"{arch}: decoder widths {ragged} are not multiples of {vec} channels")
        self._post_bx = _BNExec(post_bn)
        self._device = torch.device(device)
        self.ctx: Optional[Ctx] = None
        self._pack_tables: Dict[tuple, tuple] = {}
        self.flat_param: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self._last: Dict[str, object] = {}
        # called as hook(offset) during backward once every gradient element at flat index >= offset is final
        # (tile-DDP launches its bucketed all-reduce from here, overlapping the rest of the backward)
        self.grad_ready_hook = None
        # optional (start_event, end_event) recorded around one conv launch: {id(conv_layer): (e0, e1)}
        if self._device.type == "cuda":
            self._materialize()

    # ------------------------------------------------------------------ fastai indexing contract
    def __getitem__(self, i):
        if isinstance(i, slice):
            return nn.Sequential(*list(self.layers)[i])
        return self.layers[i]

    def __len__(self):
        return len(self.layers)

    # ------------------------------------------------------------------ flat parameter storage
    def _materialize(self):
        dev = self._device
        for b_name, b in list(self.named_buffers()):
            b.data = b.data.to(dev)
        params = list(self.parameters())
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._param_offsets = {}
        for p, o in zip(params, offs):
            view = self.flat_param[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
            self._param_offsets[id(p)] = (o, p.numel())
        self.ctx = Ctx(dev, torch.bfloat16 if self.act_dtype == "bf16" else torch.float32)
        enc = self.layers[0]
        self._enc_child_offset = {}
        for i, child in enumerate(enc):
            ps = list(child.parameters())
            if ps:
                self._enc_child_offset[i] = min(self._param_offsets[id(p)][0] for p in ps)
        self._decoder_offset = min(self._param_offsets[id(p)][0] for l in list(self.layers)[1:] for p in l.parameters())
        # readiness points of the decoder backward (tile-DDP): flat index of the first parameter of every top-level child.  The backward
        # runs from the head down, so once child i is done every gradient element at index >= _layer_offset[i] is final
        self._layer_offset = {}
        for i, l in enumerate(self.layers):
            ps = list(l.parameters())
            if i >= 1 and ps:
                self._layer_offset[i] = min(self._param_offsets[id(p)][0] for p in ps)
        for m in self.modules():
            cx = getattr(m, "cx", None)
            if isinstance(cx, _ConvExec):
                cx.ctx = self.ctx

    # ------------------------------------------------------------------ packed filter images
    def _pack_all(self, training: bool):
        """Rebuild every stale packed filter image in ONE launch (unet_pack_batch_run).  The parameters are rewritten by every
        optimizer step, so in training this runs once per step for all 52 convs x (forward, input-gradient) images.  In eval mode
        (ctx.fold_bn) the forward image of a conv that is followed by a BatchNorm is w * scale[cout] (eval coefficients, cached)."""
        cxs = self.__dict__.get("_cxs")          # (the module tree does not change after construction; walking it costs 0.5 ms of host time per step)
        if cxs is None:
            cxs = self.__dict__["_cxs"] = [m.cx for m in self.modules() if isinstance(getattr(m, "cx", None), _ConvExec)]
        jobs, marks = [], []
        for cx in cxs:
            cx.ensure_buffers(training)
            ver = cx.version()
            if cx._ver_f != ver:
                jobs.append((cx.wsrc(), cx.wp_f, 0, cx.fold_scale()))
                marks.append((cx, 0, ver))
            if training and cx._ver_d != ver:
                jobs.append((cx.wsrc(), cx.wp_d, 1, None))
                marks.append((cx, 1, ver))
            # the pixel-shuffle image of the final upsample conv (built on demand the first time, ops.conv1x1_shuffle): once it exists it is
            # rebuilt with everything else instead of by a launch of its own per step
            if cx.wp_s is not None and cx.wp_s.dtype == self.ctx.act_dtype and cx._ver_s != ver and not cx.gapped:
                jobs.append((cx.conv.weight.data, cx.wp_s, 2, None))
                marks.append((cx, 2, ver))
        if not jobs:
            return
        bf = self.ctx.act_dtype == torch.bfloat16
        side = self.ctx.side() if (training and self.ctx.overlap_now()) else None
        if side is None:
            ops.pack_jobs(jobs, bf, self._device, self._pack_tables)
        else:
            # the input-gradient images are first read when the backward starts: they are built on the weight-gradient stream next to the
            # forward pass (half of an HBM-bound launch over every parameter leaves the critical path); _hip_backward waits for the event
            fwd = [j for j in jobs if j[2] != 1]          # (modes 0 and 2: read by the forward pass)
            bwd = [j for j in jobs if j[2] == 1]
            if fwd:
                ops.pack_jobs(fwd, bf, self._device, self._pack_tables)
            if bwd:
                ready = torch.cuda.Event()
                ready.record()                      # the parameters are final on the launch stream (behind the optimizer step)
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    ops.pack_jobs(bwd, bf, self._device, self._pack_tables)
                    self.ctx.pack_d_event = torch.cuda.Event()
                    self.ctx.pack_d_event.record()
                self.ctx._side_dirty = True
        for cx, mode, ver in marks:
            if mode == 0:
                cx._ver_f = ver
            elif mode == 2:
                cx._ver_s = ver
            else:
                cx._ver_d = ver

    def param_span(self, p: nn.Parameter) -> Tuple[int, int]:
        return self._param_offsets[id(p)]

    def mark_weights_dirty(self):
        """call after parameter memory was rewritten by a HIP kernel (Adam step, all-reduce broadcast)"""
        self.ctx.weights_epoch += 1

    def zero_grad(self, set_to_none: bool = False):  # grads are views of the flat buffer: keep them
        self.flat_grad.zero_()

    def _apply(self, fn, recurse=True):
        # .cuda()/.to(device) after construction would break the flat views; the model is born on its device
        probe = fn(torch.zeros(1, device=self._device))
        if probe.device != self._device or probe.dtype != torch.float32:
            raise RuntimeError("HipDynamicUnet is bound to its device/dtype at construction (fp32 on cuda)")
        return self

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.mark_weights_dirty()
        return r

    # ------------------------------------------------------------------ programs
    def _hip_forward(self, x: torch.Tensor, training: bool) -> TS:
        """x: [B, n_in, H, W] fp32 (device).  Returns the logits slice (NHWC)."""
        if isinstance(x, ops.WindowBatch):      # windows of an integer raster / staged tiles: cut + scaled on the device (predict_raster)
            assert x.src.C == self.n_in, (x.src.C, self.n_in)
            N, H, W = x.n, x.th, x.tw
            put = x.write
        else:
            assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == self.n_in, x.shape
            x = x.contiguous()
            N, _, H, W = x.shape
            put = lambda buf, at, _x=x: ops.nchw_to_nhwc(_x, ops.TS(buf, 0, buf.shape[3]), at=at)
        ctx = self.ctx
        ctx.training = training
        ctx.main_stream = None
        ctx.step_pixels = N * H * W
        ctx.fold_bn = bool(self.fold_eval_bn) and not training     # eval: Conv + BN + ReLU = ONE launch (BatchNorm folded into filter + bias)
        self._pack_all(training and ctx.need_grad)
        L = self.layers
        enc: Encoder = L[0]
        x0 = ctx.act(self, "x0", N, H, W, self.n_in, zero=True)
        put(x0.buf, 0)
        skips: Dict[int, TS] = {}
        h = x0
        for i, child in enumerate(enc):
            if i == 3:
                OH, OW = (h.H + 2 - 3) // 2 + 1, (h.W + 2 - 3) // 2 + 1
                y = ctx.act(child, "y", N, OH, OW, h.C)
                idx = ctx.vec(child, f"idx{N}x{OH}x{OW}", N * OH * OW * h.C, dtype=torch.uint8)
                ops.maxpool(h, y, idx)
                ctx.saved[(id(child), "x")] = h
                h = y
            elif i < 3:
                h = child.hip_fwd(ctx, h)
            else:
                for blk in child:
                    h = blk.hip_fwd(ctx, h)
            if i in self.sz_chg_idxs:
                skips[i] = h
        # BatchNorm(ni) -> ReLU -> middle_conv
        e = h
        scale, shift = self._post_bx.coeffs(ctx, e)
        m0 = ctx.act(self, "m0", e.N, e.H, e.W, e.C)
        ops.affine_act(e, m0, scale, shift, relu=True)
        ctx.saved[(id(self), "e")] = e
        h = m0
        for cl in L[3]:
            a = ctx.act(cl, "a", h.N, h.H, h.W, cl.nf)
            cl.cx.fwd(h, a, relu=True)
            ctx.saved[(id(cl), "x")] = h
            h = a
        for k, idx in enumerate(self.sz_chg_idxs):
            h = L[4 + k].hip_fwd(ctx, h, skips[idx])
        nb = 4 + len(self.sz_chg_idxs)
        X = ctx.act(self, "xcat", N, H, W, self.cat_p, zero=True)
        # the network input behind the up-sampled channels: appended by the fused upsample launch from x0 (the same values, already NHWC) where
        # it takes that; else a second pass -- after the shuffle: the up slice owns its padding lanes (a gap: zeros) in front of these channels
        if not L[nb].hip_fwd(ctx, h, X.sub(0, self.up_c), (H, W), tail=(x0, self.up_off)):
            put(X.buf, self.up_off)
        o = L[nb + 3].hip_fwd(ctx, X)
        z = ctx.act(self, "logits", N, H, W, self.n_out, zero=True, dtype=torch.float32)      # logits are fp32 in both modes
        head: ConvLayer = L[nb + 4]
        head.cx.fwd(o, z)
        ctx.saved[(id(head), "x")] = o
        self._last = {"skips": skips, "N": N, "H": H, "W": W, "z": z, "training": training}
        return z

    def _hip_backward(self, dz: TS):
        """dz: dL/dlogits (NHWC slice, same geometry as the logits).  Fills every .grad view."""
        ctx = self.ctx
        L = self.layers
        enc: Encoder = L[0]
        last = self._last
        assert last.get("training", False), "backward needs a preceding training-mode forward"
        ctx.main_stream = torch.cuda.current_stream()           # (once per backward: every event of the program is recorded on / waited for by this object)
        if getattr(ctx, "pack_d_event", None) is not None:      # input-gradient filter images built on the second stream during the forward
            ctx.main_stream.wait_event(ctx.pack_d_event)
            ctx.pack_d_event = None
        skips: Dict[int, TS] = last["skips"]
        nb = 4 + len(self.sz_chg_idxs)
        head: ConvLayer = L[nb + 4]
        o: TS = ctx.saved[(id(head), "x")]
        do = head.bwd_from_dy(ctx, dz, mask=o)                  # masked by relu of the final ResBlock
        dX = L[nb + 3].bwd_nonorm(ctx, do, dx_channels=self.up_c)    # no gradient for the network-input channels of the concat
        ctx.free(do)
        d = L[nb].hip_bwd(ctx, dX.sub(0, self.up_c))            # -> masked grad wrt UnetBlock 3 conv2 pre-activation
        ctx.free(dX)
        hook = self.grad_ready_hook
        if hook is not None:
            user_hook = hook

            def hook(offset, _h=user_hook, _c=ctx):             # the weight gradients up to here may still run on the side stream
                _c.side_join()
                _h(offset)
        if hook is not None:                                    # head, final ResBlock, final shuffle: the first bucket can leave now
            hook(self._layer_offset[nb])
        dskips: Dict[int, TS] = {}
        for k in range(len(self.sz_chg_idxs) - 1, -1, -1):
            idx = self.sz_chg_idxs[k]
            s = skips[idx]
            ds = ctx.tmp(s.N, s.H, s.W, s.C)                    # lives until the encoder backward reaches child idx
            dn = L[4 + k].hip_bwd(ctx, d, ds, False)
            ctx.free(d)
            d = dn
            dskips[idx] = ds
            if hook is not None:                                # one readiness point per UnetBlock
                hook(self._layer_offset[4 + k])
        # middle_conv (d is masked wrt middle_conv[1] pre-activation)
        mids = list(L[3])
        for j in range(len(mids) - 1, -1, -1):
            xin: TS = ctx.saved[(id(mids[j]), "x")]
            dn = mids[j].bwd_from_dy(ctx, d, mask=xin)          # xin is a ReLU output in both cases (m0, mid0 out)
            ctx.free(d)
            d = dn
        # post-encoder BN (ReLU already applied through the mask above)
        e: TS = ctx.saved[(id(self), "e")]
        de = ctx.tmp(e.N, e.H, e.W, e.C)
        self._post_bx.bwd(ctx, d, None, e, de)
        ctx.free(d)
        if hook is not None:
            hook(self._decoder_offset)
        # encoder, last child to first
        d = de
        children = list(enc)
        for i in range(len(children) - 1, -1, -1):
            child = children[i]
            if i in dskips and i != len(children) - 1:
                ops.copy_slice(dskips[i], d, accumulate=True)   # skip gradient joins the encoder gradient
                ctx.free(dskips.pop(i))
            if i > 3:
                for blk in reversed(list(child)):
                    dn = blk.hip_bwd(ctx, d)
                    ctx.free(d)
                    d = dn
            elif i == 3:
                xin: TS = ctx.saved[(id(child), "x")]
                idxb = ctx.vec(child, f"idx{d.N}x{d.H}x{d.W}", d.N * d.H * d.W * d.C, dtype=torch.uint8)
                dx = ctx.tmp(xin.N, xin.H, xin.W, xin.C)
                ops.maxpool_bwd(d, idxb, dx)
                ctx.free(d)
                d = dx
            else:
                dn = child.hip_bwd(ctx, d, need_dx=(i != 0))
                ctx.free(d)
                d = dn
            if hook is not None and i in self._enc_child_offset:
                hook(self._enc_child_offset[i])
        for t in dskips.values():
            ctx.free(t)
        ctx.side_join()                                         # every .grad view is final behind this point of the launch stream
        ctx.main_stream = None
        assert not ctx._pool_live, "a backward temporary was not returned to the pool"

    # ------------------------------------------------------------------ torch-facing surface
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Raw logits [B, n_out, H, W] (a channels-last view of the NHWC logits buffer, no copy).
        With grad enabled in training mode the result carries an autograd node that runs the HIP backward."""
        if self.flat_param is None:
            raise RuntimeError("HipDynamicUnet needs a GPU: it has no CPU fallback")
        x = x.to(self._device, torch.float32)
        if self.training and torch.is_grad_enabled():
            return _UnetFunction.apply(self, x, self.flat_param.requires_grad_(True))
        z = self._hip_forward(x, self.training)
        return z.view().permute(0, 3, 1, 2)

    @torch.no_grad()
    def predict_probs_graphed(self, x: torch.Tensor):
        """predict_probs through a hipGraph captured per input geometry: the per-tile loop of the reference (batch 1,
        predict.py:191-193) is launch-bound, a replay is one submission."""
        x = x.to(self._device, torch.float32)
        key = tuple(x.shape)
        cache = self.__dict__.setdefault("_pred_graphs", {})
        ent = cache.get(key)
        if ent is None:
            self.predict_probs(x); self.predict_probs(x)            # warm-up allocates the persistent buffers
            xs = x.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                probs, amax = self.predict_probs(xs)
            ent = cache[key] = (g, xs, probs, amax, (self.ctx.weights_epoch, self.ctx.bn_epoch))
        g, xs, probs, amax, epoch = ent
        if epoch != (self.ctx.weights_epoch, self.ctx.bn_epoch):     # weights or running statistics changed: the captured filter images / shifts are stale
            del cache[key]
            return self.predict_probs_graphed(x)
        xs.copy_(x, non_blocking=True)
        g.replay()
        return probs, amax

    def logits_ts(self) -> TS:
        return self._last["z"]

    def _ensure_grad_views(self):
        """torch.optim's zero_grad(set_to_none=True) drops .grad: re-attach the flat-buffer views."""
        for p in self.parameters():
            o, n = self._param_offsets[id(p)]
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def forward_loss_backward(self, x: torch.Tensor, y: torch.Tensor, weight: Optional[torch.Tensor] = None,
                              grad_scale: float = 1.0, reg_kind: Optional[str] = None, reg_beta: float = 0.5,
                              world: int = 1, focal_gamma: Optional[float] = None) -> torch.Tensor:
        """One fused training pass: logits -> loss -> backward into the flat gradient buffer.  Returns the loss as a 1-element
        device tensor (no host sync).  grad_scale multiplies the gradient.
        world > 1 (tile-DDP): the weighted cross-entropy is sum_r num_r / sum_r den_r over the ranks (den_r = sum of w[y] on
        rank r differs per rank when the class weights are not uniform): numerator and denominator are
        all-reduced between the loss forward and backward kernels, the local gradient is taken w.r.t. the GLOBAL denominator
        and the SUM all-reduce of the gradients then equals the single-process gradient of the global batch.  Regression
        losses are plain means over equally many pixels per rank: gradient pre-scaled by 1/world.
        Classification (default): weighted per-pixel cross-entropy, CrossEntropyLossFlat(axis=1, weight) (train.py:195,211).
        focal_gamma: FocalLossFlat(gamma, axis=1) instead (params_and_main.py:87-89): a plain mean over equally many pixels per rank, like the
        regression losses -- loss averaged over the ranks, gradient pre-scaled by 1 / world.
        Regression (reg_kind = "mse" | "l1" | "smoothl1", n_out = 1, float targets [B,H,W]): train.py:189-193."""
        x = x.to(self._device, torch.float32)
        z = self._hip_forward(x, True)
        ctx = self.ctx
        P = z.P
        dz = ctx.act(self, "dlogits", z.N, z.H, z.W, z.C, zero=True)
        if reg_kind is None and focal_gamma is not None:
            y = y.to(self._device, torch.int64).contiguous()
            loss = ctx.vec(self, "loss", 1)
            ops.focal_fwd(z, y, weight, focal_gamma, loss, ctx.workspace(ops.ce_workspace(P)))
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(loss)
                loss.div_(world)
                grad_scale = grad_scale / world
            ops.focal_bwd(z, y, weight, focal_gamma, grad_scale, dz)
        elif reg_kind is None:
            y = y.to(self._device, torch.int64).contiguous()
            loss, denom = ctx.vec(self, "loss", 1), ctx.vec(self, "denom", 1)
            if world > 1:
                import torch.distributed as dist
                nd = ctx.vec(self, "numden", 2)
                ops.ce_fwd_parts(z, y, weight, nd, ctx.workspace(ops.ce_workspace(P)))      # this rank's numerator and denominator
                dist.all_reduce(nd)
                denom.copy_(nd[1:2])
                torch.div(nd[0:1], nd[1:2], out=loss)
            else:
                ops.ce_fwd(z, y, weight, loss, denom, ctx.workspace(ops.ce_workspace(P)))
            ops.ce_bwd(z, y, weight, denom, grad_scale, dz)
        else:
            if self.n_out != 1:
                raise ValueError("regression losses need a model with n_out = 1")
            y = y.to(self._device, torch.float32).contiguous()
            loss = ctx.vec(self, "loss", 1)
            ops.regloss_fwd(z, y, reg_kind, reg_beta, loss, ctx.workspace(ops.ce_workspace(P)))
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(loss)
                loss.div_(world)
                grad_scale = grad_scale / world
            if dz.bf16:          # the loss kernels write fp32 (the logits are fp32 in both modes): one cast into the bf16 gradient slice
                dz32 = ctx.act(self, "dlogits32", z.N, z.H, z.W, z.C, zero=True, dtype=torch.float32)
                ops.regloss_bwd(z, y, reg_kind, reg_beta, grad_scale, dz32)
                ops.cast_slice(dz32, dz)
            else:
                ops.regloss_bwd(z, y, reg_kind, reg_beta, grad_scale, dz)
        self._ensure_grad_views()
        self._hip_backward(dz)
        return loss

    @torch.no_grad()
    def predict_values(self, x: torch.Tensor) -> torch.Tensor:
        """eval-mode forward, raw outputs [B,n_out,H,W] (regression: no activation, train.py:90-95)."""
        x = x.to(self._device, torch.float32)
        z = self._hip_forward(x, False)
        out = torch.empty((z.N, z.C, z.H, z.W), dtype=torch.float32, device=self._device)
        ops.nhwc_to_nchw(z, out)
        return out

    @torch.no_grad()
    def predict_probs(self, x: torch.Tensor, want_probs=True, want_argmax=True):
        """eval-mode forward + softmax(dim=1) + argmax: what Learner.predict returns per tile (predict.py:193-203,232)."""
        x = x.to(self._device, torch.float32)
        z = self._hip_forward(x, False)
        probs = torch.empty((z.N, z.C, z.H, z.W), dtype=torch.float32, device=self._device) if want_probs else None
        amax = torch.empty((z.N, z.H, z.W), dtype=torch.int64, device=self._device) if want_argmax else None
        ops.softmax_argmax(z, probs, amax)
        return probs, amax

    @torch.no_grad()
    def forward_windows(self, wb: "ops.WindowBatch") -> TS:
        """eval-mode forward of a batch of raster windows; returns the fp32 NHWC logits slice (valid until the next forward)"""
        return self._hip_forward(wb, False)

    def memory_bytes(self) -> int:
        return self.ctx.bytes_allocated() + 2 * self.flat_param.numel() * 4


class _UnetFunction(torch.autograd.Function):
    """Bridges torch autograd (loss.backward() in a fastai-style loop) to the HIP backward program.
    Parameter gradients are written straight into the flat gradient buffer (the .grad views)."""

    @staticmethod
    def forward(ctx, model: HipDynamicUnet, x: torch.Tensor, flat_param: torch.Tensor):
        z = model._hip_forward(x, True)
        ctx.model = model
        return z.view().permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        model: HipDynamicUnet = ctx.model
        z: TS = model._last["z"]
        dz = model.ctx.act(model, "dlogits", z.N, z.H, z.W, z.C, zero=True)
        dz.view().copy_(dlogits.permute(0, 2, 3, 1))
        model._ensure_grad_views()
        model._hip_backward(dz)
        return None, None, None
