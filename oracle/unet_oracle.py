"""CPU ORACLE -- test infrastructure only, never the product path.

A plain-PyTorch (CPU, fp32, NCHW) restatement of the network, loss, metric and
optimiser step that the reference builds through ``fastai==2.5.1``:

* reference call sites: ``train.py:98-160`` (``unet_learner_MS``: xresnet body,
  stem swap, ``DynamicUnet(... blur=True, blur_final=True, self_attention,
  y_range=None, norm_type=NormType, last_cross=True, bottle=False)``),
  ``train.py:163-250`` (weighted ``CrossEntropyLossFlat(axis=1)``, ``DiceMulti``,
  ``Adam``, ``fit_one_cycle(lr_max=slice(lr/f, lr))``), ``train.py:78-80``
  (3 parameter groups), ``predict.py:191-203,232`` (softmax probs -> argmax).
* the arithmetic itself lives in the un-vendored dependency fastai 2.5.1
  (``environment/requirements.txt:4``): ``vision/models/unet.py``,
  ``vision/models/xresnet.py``, ``layers.py``, ``losses.py``, ``metrics.py``,
  ``optimizer.py``, ``callback/schedule.py``.  fastai is absent from this image,
  so these modules are restated from its published behaviour.

PARITY UNPINNED: the reference repository holds no tests, fixtures or golden
vectors for this path and fastai cannot be imported here, so this oracle is
pinned only by review against SURVEY.md section 8(a) plus structural anchors
(41 244 577 parameters for xresnet34 4->5, skip indices [2,4,5,6], fastai
state-dict key scheme).  The HIP path is pinned to THIS oracle by tests/.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# layers.py restatement
# --------------------------------------------------------------------------

BN_EPS = 1e-5
BN_MOM = 0.1


def batchnorm(nf: int, zero: bool = False) -> nn.BatchNorm2d:
    """fastai ``BatchNorm``: bias 1e-3, weight 1 (0 for NormType.BatchZero)."""
    bn = nn.BatchNorm2d(nf, eps=BN_EPS, momentum=BN_MOM)
    with torch.no_grad():
        bn.bias.fill_(1e-3)
        bn.weight.fill_(0.0 if zero else 1.0)
    return bn


class ConvLayer(nn.Sequential):
    """fastai ``ConvLayer`` for the norm types this path uses.

    ``norm``: 'batch' | 'batchzero' -> conv(no bias) + BN [+ ReLU]   (encoder)
              None                  -> conv(bias)   [+ ReLU]         (decoder: the
              reference passes the enum *class* NormType, train.py:100,142, which
              matches no branch, so there is no norm and the conv has a bias).
    Order is conv, BN, act (bn_1st=True); ``xtra`` (self-attention) goes last.
    """

    def __init__(self, ni, nf, ks=3, stride=1, norm: Optional[str] = "batch", act=True,
                 bias_std=0.01, xtra: Optional[nn.Module] = None):
        bn = norm in ("batch", "batchzero")
        conv = nn.Conv2d(ni, nf, ks, stride=stride, padding=(ks - 1) // 2, bias=not bn)
        # init_linear(..., init='auto'): kaiming_uniform_ when followed by ReLU
        with torch.no_grad():
            if conv.bias is not None:
                if bias_std != 0:
                    conv.bias.normal_(0, bias_std)
                else:
                    conv.bias.zero_()
            if act:
                nn.init.kaiming_uniform_(conv.weight)
        layers: List[nn.Module] = [conv]
        if bn:
            layers.append(batchnorm(nf, zero=(norm == "batchzero")))
        if act:
            layers.append(nn.ReLU())
        if xtra is not None:
            layers.append(xtra)
        super().__init__(*layers)


class ResBlock(nn.Module):
    """fastai ``ResBlock`` (expansion 1 or 4), AvgPool(2, ceil_mode=True) first on
    a strided identity path, 1x1 ConvLayer(+BN, no act) when widths differ."""

    def __init__(self, expansion, ni, nf, stride=1, norm: Optional[str] = "batch"):
        super().__init__()
        norm2 = "batchzero" if norm == "batch" else norm
        nh = nf
        nf, ni = nf * expansion, ni * expansion
        if expansion == 1:
            convpath = [ConvLayer(ni, nh, 3, stride=stride, norm=norm),
                        ConvLayer(nh, nf, 3, norm=norm2, act=False)]
        else:
            convpath = [ConvLayer(ni, nh, 1, norm=norm),
                        ConvLayer(nh, nh, 3, stride=stride, norm=norm),
                        ConvLayer(nh, nf, 1, norm=norm2, act=False)]
        self.convpath = nn.Sequential(*convpath)
        idpath: List[nn.Module] = []
        if ni != nf:
            # NB: the identity ConvLayer keeps fastai's default NormType.Batch even
            # inside a norm-free decoder ResBlock (kwargs carry no norm_type).
            idpath.append(ConvLayer(ni, nf, 1, norm="batch", act=False))
        if stride != 1:
            idpath.insert(0, nn.AvgPool2d(stride, ceil_mode=True))
        self.idpath = nn.Sequential(*idpath)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.act(self.convpath(x) + self.idpath(x))


def icnr_init(x: torch.Tensor, scale=2, init=nn.init.kaiming_normal_) -> torch.Tensor:
    """ICNR: every group of scale**2 output channels starts identical."""
    ni, nf, h, w = x.shape
    ni2 = int(ni / (scale ** 2))
    k = init(x.new_zeros([ni2, nf, h, w])).transpose(0, 1)
    k = k.contiguous().view(ni2, nf, -1)
    k = k.repeat(1, 1, scale ** 2)
    return k.contiguous().view([nf, ni, h, w]).transpose(0, 1)


class PixelShuffleICNR(nn.Sequential):
    """fastai ``PixelShuffle_ICNR``: 1x1 ConvLayer(ni -> 4 nf, bias, ReLU) ->
    PixelShuffle(2) [-> ReplicationPad2d((1,0,1,0)) -> AvgPool2d(2, stride=1)]."""

    def __init__(self, ni, nf=None, blur=False):
        nf = ni if nf is None else nf
        layers: List[nn.Module] = [ConvLayer(ni, nf * 4, ks=1, norm=None, bias_std=0), nn.PixelShuffle(2)]
        with torch.no_grad():
            layers[0][0].weight.copy_(icnr_init(layers[0][0].weight.data))
        if blur:
            layers += [nn.ReplicationPad2d((1, 0, 1, 0)), nn.AvgPool2d(2, stride=1)]
        super().__init__(*layers)


class SelfAttention(nn.Module):
    """fastai ``SelfAttention`` (SAGAN style, softmax over dim=1, gamma init 0,
    legacy spectral norm on the three 1x1 Conv1d projections)."""

    def __init__(self, n_channels):
        super().__init__()
        self.query = self._conv(n_channels, n_channels // 8)
        self.key = self._conv(n_channels, n_channels // 8)
        self.value = self._conv(n_channels, n_channels)
        self.gamma = nn.Parameter(torch.tensor([0.0]))

    @staticmethod
    def _conv(n_in, n_out):
        conv = nn.Conv1d(n_in, n_out, 1, bias=False)
        return nn.Sequential(nn.utils.spectral_norm(conv))

    def forward(self, x):
        size = x.size()
        x = x.view(*size[:2], -1)
        f, g, h = self.query(x), self.key(x), self.value(x)
        beta = F.softmax(torch.bmm(f.transpose(1, 2), g), dim=1)
        o = self.gamma * torch.bmm(h, beta) + x
        return o.view(*size).contiguous()


# --------------------------------------------------------------------------
# vision/models/xresnet.py restatement (body only: create_body cuts at the pool)
# --------------------------------------------------------------------------

# every constructor the reference imports (params_and_main.py:12): expansion, blocks per stage.  fastai XResNet: stage widths
# [64, 128, 256, 512] + [256] * (len(layers) - 4), every stage after the first has stride 2 (xresnet34_deep: two more halvings)
XRESNET_LAYERS = {"xresnet18": (1, [2, 2, 2, 2]), "xresnet34": (1, [3, 4, 6, 3]), "xresnet50": (4, [3, 4, 6, 3]),
                  "xresnet101": (4, [3, 4, 23, 3]), "xresnet34_deep": (1, [3, 4, 6, 3, 1, 1])}


def init_cnn(m: nn.Module):
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0)
    if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
        nn.init.kaiming_normal_(m.weight)
    for l in m.children():
        init_cnn(l)


def xresnet_body(arch: str, c_in: int) -> nn.Sequential:
    """Children of fastai's XResNet up to the pooling layer: 3 stem ConvLayers (32,32,64; first is
    stride 2), MaxPool2d(3,2,1), the 4 (xresnet34_deep: 6) stages of ResBlocks.  The reference replaces
    the first conv by a fresh ``nn.Conv2d(c_in, 32, 3, 2, 1, bias=False)`` with
    PyTorch's default init (train.py:130-135), reproduced here."""
    expansion, layers = XRESNET_LAYERS[arch]
    stem_szs = [3, 32, 32, 64]
    stem = [ConvLayer(stem_szs[i], stem_szs[i + 1], 3, stride=2 if i == 0 else 1) for i in range(3)]
    block_szs = [64 // expansion, 64, 128, 256, 512] + [256] * (len(layers) - 4)
    stages = []
    for i, nb in enumerate(layers):
        ni, nf = block_szs[i], block_szs[i + 1]
        stages.append(nn.Sequential(*[
            ResBlock(expansion, ni if j == 0 else nf, nf, stride=(1 if i == 0 else 2) if j == 0 else 1)
            for j in range(nb)]))
    body = nn.Sequential(*stem, nn.MaxPool2d(3, stride=2, padding=1), *stages)
    init_cnn(body)
    body[0][0] = nn.Conv2d(c_in, 32, kernel_size=3, stride=2, padding=1, bias=False)
    return body


# --------------------------------------------------------------------------
# vision/models/unet.py restatement
# --------------------------------------------------------------------------

def _apply_init_kaiming(m: nn.Module):
    """fastai ``apply_init(m, kaiming_normal_)``: weights of non-norm layers with
    dim>1 re-drawn, their biases zeroed."""
    for l in m.modules():
        if isinstance(l, (nn.BatchNorm1d, nn.BatchNorm2d)):
            continue
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            if hasattr(l, "weight") and isinstance(l.weight, torch.Tensor) and l.weight.dim() > 1:
                nn.init.kaiming_normal_(l.weight)
            if getattr(l, "bias", None) is not None:
                with torch.no_grad():
                    l.bias.fill_(0.0)


class UnetBlock(nn.Module):
    def __init__(self, up_in_c, x_in_c, final_div=True, blur=False, self_attention=False):
        super().__init__()
        self.shuf = PixelShuffleICNR(up_in_c, up_in_c // 2, blur=blur)
        self.bn = batchnorm(x_in_c)
        ni = up_in_c // 2 + x_in_c
        nf = ni if final_div else ni // 2
        self.conv1 = ConvLayer(ni, nf, norm=None)
        self.conv2 = ConvLayer(nf, nf, norm=None, xtra=SelfAttention(nf) if self_attention else None)
        self.relu = nn.ReLU()
        _apply_init_kaiming(nn.Sequential(self.conv1, self.conv2))
        self.out_channels = nf

    def forward(self, up_in, s):
        up_out = self.shuf(up_in)
        if s.shape[-2:] != up_out.shape[-2:]:
            up_out = F.interpolate(up_out, s.shape[-2:], mode="nearest")
        cat_x = self.relu(torch.cat([up_out, self.bn(s)], dim=1))
        return self.conv2(self.conv1(cat_x))


class DynamicUnet(nn.Module):
    """fastai ``DynamicUnet`` as the reference configures it.  ``self.layers`` is
    indexed exactly like fastai's SequentialEx so state-dict keys coincide:
    0 encoder, 1 BN, 2 ReLU, 3 middle_conv, 4-7 UnetBlocks, 8 PixelShuffle_ICNR,
    9 ResizeToOrig, 10 MergeLayer(dense), 11 ResBlock, 12 head ConvLayer."""

    def __init__(self, arch: str, n_in: int, n_out: int, img_size=(64, 64), self_attention=False):
        super().__init__()
        enc = xresnet_body(arch, n_in)
        with torch.no_grad():
            enc.eval()
            x = torch.zeros(1, n_in, *img_size)
            sizes = []
            for child in enc:
                x = child(x)
                sizes.append(tuple(x.shape))
            enc.train()
        fs = [s[-1] for s in sizes]
        self.sz_chg_idxs = [i for i in range(len(fs) - 1) if fs[i] != fs[i + 1]][::-1]
        ni = sizes[-1][1]
        middle = nn.Sequential(ConvLayer(ni, ni * 2, norm=None), ConvLayer(ni * 2, ni, norm=None))
        layers: List[nn.Module] = [enc, batchnorm(ni), nn.ReLU(), middle]
        xc = ni
        for i, idx in enumerate(self.sz_chg_idxs):
            not_final = i != len(self.sz_chg_idxs) - 1
            sa = self_attention and (i == len(self.sz_chg_idxs) - 3)
            blk = UnetBlock(xc, sizes[idx][1], final_div=not_final, blur=True, self_attention=sa)
            layers.append(blk)
            xc = blk.out_channels
        layers.append(PixelShuffleICNR(xc))          # imsize != sizes[0][-2:] always (stem is stride 2)
        layers.append(nn.Identity())                 # ResizeToOrig (functional, see forward)
        layers.append(nn.Identity())                 # MergeLayer(dense=True)
        xc += n_in
        layers.append(ResBlock(1, xc, xc, norm=None))
        layers.append(ConvLayer(xc, n_out, ks=1, norm=None, act=False))
        _apply_init_kaiming(nn.Sequential(layers[3], layers[-2]))
        self.layers = nn.ModuleList(layers)
        self.n_in, self.n_out, self.arch = n_in, n_out, arch

    # fastai indexing contract used by the splitter (train.py:78-80)
    def __getitem__(self, i):
        if isinstance(i, slice):
            return nn.Sequential(*list(self.layers)[i])
        return self.layers[i]

    def forward(self, x, taps: Optional[Dict[str, torch.Tensor]] = None):
        orig = x
        skips = {}
        h = x
        for i, child in enumerate(self.layers[0]):
            h = child(h)
            if i in self.sz_chg_idxs:
                skips[i] = h
        if taps is not None:
            taps["encoder"] = h
        h = self.layers[2](self.layers[1](h))
        h = self.layers[3](h)
        if taps is not None:
            taps["middle"] = h
        for k, idx in enumerate(self.sz_chg_idxs):
            h = self.layers[4 + k](h, skips[idx])
            if taps is not None:
                taps[f"unet{k}"] = h
        nb = 4 + len(self.sz_chg_idxs)
        h = self.layers[nb](h)
        if h.shape[-2:] != orig.shape[-2:]:
            h = F.interpolate(h, orig.shape[-2:], mode="nearest")
        h = torch.cat([h, orig], dim=1)
        h = self.layers[nb + 3](h)
        if taps is not None:
            taps["final_res"] = h
        return self.layers[nb + 4](h)


def xresnet_split(m: DynamicUnet) -> List[List[nn.Parameter]]:
    """``_xresnet_split`` (train.py:78-80): stem / rest of encoder / decoder."""
    enc = m.layers[0]
    g0 = [p for l in list(enc)[:3] for p in l.parameters()]
    g1 = [p for l in list(enc)[3:] for p in l.parameters()]
    g2 = [p for l in list(m.layers)[1:] for p in l.parameters()]
    return [g0, g1, g2]


def bn_bias_params(m: nn.Module) -> List[nn.Parameter]:
    """fastai ``norm_bias_params(model, with_bias=True)``: every parameter of a
    norm layer plus the bias of every other layer -- these get no weight decay
    when ``wd_bn_bias=False`` (train.py:102,152-154)."""
    out: List[nn.Parameter] = []
    for l in m.modules():
        if isinstance(l, (nn.BatchNorm1d, nn.BatchNorm2d)):
            out += list(l.parameters(recurse=False))
        elif getattr(l, "bias", None) is not None and isinstance(l.bias, nn.Parameter):
            out.append(l.bias)
    return out


# --------------------------------------------------------------------------
# losses.py / metrics.py restatement
# --------------------------------------------------------------------------

class CrossEntropyLossFlat:
    """``CrossEntropyLossFlat(axis=1, weight=w)``; ``.func.weight`` is assigned by
    the reference (train.py:211); ``activation`` = softmax(dim=1), ``decodes`` =
    argmax(dim=1) are what ``Learner.predict`` uses (predict.py:193)."""

    def __init__(self, weight: Optional[torch.Tensor] = None, axis: int = 1):
        self.func = nn.CrossEntropyLoss(weight=weight)
        self.axis = axis

    def __call__(self, inp: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        # fastai BaseLoss.__call__: `inp, targ = map(self._contiguous, (inp, targ))` -- BOTH are
        # transposed (axis <-> last) before flattening, so pixel i of inp meets pixel i of targ.
        inp = inp.transpose(self.axis, -1).contiguous()
        targ = targ.transpose(self.axis, -1).contiguous()
        return self.func(inp.view(-1, inp.shape[-1]), targ.view(-1))

    def activation(self, x):
        return F.softmax(x, dim=self.axis)

    def decodes(self, x):
        return x.argmax(dim=self.axis)


class FocalLossFlat:
    """fastai 2.5.1 ``FocalLossFlat(gamma, axis=1)`` = ``BaseLoss(FocalLoss, gamma=gamma, axis=axis)`` -- the alternative classification
    loss of the reference's configuration (params_and_main.py:87-89).  ``FocalLoss.forward``: ``ce = F.cross_entropy(inp, targ,
    weight=self.weight, reduction="none"); p_t = exp(-ce); loss = ((1 - p_t) ** gamma * ce).mean()``; ``.func.weight`` is assigned by
    the reference for every loss (train.py:211)."""

    class _Func:
        def __init__(self, gamma, weight):
            self.gamma, self.weight = gamma, weight

    def __init__(self, gamma: float = 2.0, weight: Optional[torch.Tensor] = None, axis: int = 1):
        self.func = FocalLossFlat._Func(gamma, weight)
        self.axis = axis

    def __call__(self, inp: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        inp = inp.transpose(self.axis, -1).contiguous()          # BaseLoss.__call__: input AND target transposed, then flattened
        targ = targ.transpose(self.axis, -1).contiguous()
        w = self.func.weight
        ce = F.cross_entropy(inp.view(-1, inp.shape[-1]), targ.view(-1), weight=None if w is None else w.to(inp.dtype), reduction="none")
        p_t = torch.exp(-ce)
        return ((1 - p_t) ** self.func.gamma * ce).mean()

    def activation(self, x):
        return F.softmax(x, dim=self.axis)

    def decodes(self, x):
        return x.argmax(dim=self.axis)


class _FlatRegLoss:
    """fastai ``BaseLoss(loss_cls, axis=1, floatify=True, is_2d=False)`` as used by the regression branch
    (train.py:189-193 ``MSELossFlat(axis=1)``; utils.py:145-147 ``Smoothl1``): both tensors are transposed
    (axis <-> last), the target is cast to float, both are flattened with ``view(-1)``; 'mean' reduction."""
    loss_cls, kwargs = nn.MSELoss, {}

    def __init__(self, axis: int = 1):
        self.func = self.loss_cls(**self.kwargs)
        self.axis = axis

    def __call__(self, inp: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        inp = inp.transpose(self.axis, -1).contiguous()
        targ = targ.transpose(self.axis, -1).contiguous().float()
        return self.func(inp.view(-1), targ.view(-1))

    def activation(self, x):
        return x

    def decodes(self, x):
        return x


class MSELossFlat(_FlatRegLoss):
    loss_cls = nn.MSELoss


class L1LossFlat(_FlatRegLoss):
    loss_cls = nn.L1Loss


class Smoothl1(_FlatRegLoss):
    loss_cls, kwargs = nn.SmoothL1Loss, {"beta": 0.5}


def rmse(pred: torch.Tensor, targ: torch.Tensor) -> float:
    """fastai ``rmse`` (AccumMetric over the whole validation set): sqrt(mse(flatten(pred), flatten(targ)))."""
    return float(torch.sqrt(F.mse_loss(pred.reshape(-1).double(), targ.reshape(-1).double())))


def r2_score(pred: torch.Tensor, targ: torch.Tensor) -> float:
    """fastai ``R2Score()`` = sklearn.metrics.r2_score(targ, pred): 1 - SS_res / SS_tot."""
    p, t = pred.reshape(-1).double(), targ.reshape(-1).double()
    return float(1.0 - ((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())


class DiceMulti:
    """fastai ``DiceMulti(axis=1)``: per class inter/union accumulated over the
    whole validation set, value = nanmean_c(2 inter / union)."""

    def __init__(self, axis=1):
        self.axis = axis
        self.reset()

    def reset(self):
        self.inter: Dict[int, float] = {}
        self.union: Dict[int, float] = {}

    def accumulate(self, pred: torch.Tensor, targ: torch.Tensor):
        n_cls = pred.shape[self.axis]
        p_ = pred.argmax(dim=self.axis).reshape(-1)
        t_ = targ.reshape(-1)
        for c in range(n_cls):
            p = (p_ == c).to(torch.float32)
            t = (t_ == c).to(torch.float32)
            self.inter[c] = self.inter.get(c, 0.0) + float((p * t).sum())
            self.union[c] = self.union.get(c, 0.0) + float((p + t).sum())

    @property
    def value(self) -> float:
        scores = [2.0 * self.inter[c] / self.union[c] if self.union[c] > 0 else np.nan for c in self.inter]
        return float(np.nanmean(np.array(scores)))


# --------------------------------------------------------------------------
# optimizer.py / callback/schedule.py restatement
# --------------------------------------------------------------------------

class FastaiAdam:
    """fastai ``Adam(mom=.9, sqr_mom=.99, eps=1e-5, wd=0.01, decouple_wd=True)``
    over parameter groups with per-group lr; ``no_wd`` holds the parameters that
    skip weight decay (``wd_bn_bias=False``)."""

    def __init__(self, groups: Sequence[Sequence[nn.Parameter]], lr, mom=0.9, sqr_mom=0.99, eps=1e-5, wd=0.01,
                 no_wd: Sequence[nn.Parameter] = ()):
        self.groups = [list(g) for g in groups]
        self.lrs = [lr] * len(self.groups) if np.isscalar(lr) else list(lr)
        self.mom, self.sqr_mom, self.eps, self.wd = mom, sqr_mom, eps, wd
        self.no_wd = {id(p) for p in no_wd}
        self.state: Dict[int, dict] = {}

    @torch.no_grad()
    def step(self):
        for g, lr in zip(self.groups, self.lrs):
            for p in g:
                if p.grad is None:
                    continue
                st = self.state.setdefault(id(p), {"step": 0, "grad_avg": torch.zeros_like(p), "sqr_avg": torch.zeros_like(p)})
                if id(p) not in self.no_wd and self.wd != 0:
                    p.mul_(1 - lr * self.wd)
                st["grad_avg"].mul_(self.mom).add_(p.grad, alpha=1 - self.mom)
                st["sqr_avg"].mul_(self.sqr_mom).addcmul_(p.grad, p.grad, value=1 - self.sqr_mom)
                st["step"] += 1
                debias1 = 1 - self.mom ** st["step"]
                debias2 = 1 - self.sqr_mom ** st["step"]
                p.addcdiv_(st["grad_avg"], (st["sqr_avg"] / debias2).sqrt() + self.eps, value=-lr / debias1)

    def zero_grad(self):
        for g in self.groups:
            for p in g:
                p.grad = None


def even_mults(start: float, stop: float, n: int) -> np.ndarray:
    """fastai ``even_mults``: geometric progression start..stop with n entries."""
    if n == 1:
        return np.array([stop])
    mult = stop / start
    step = mult ** (1 / (n - 1))
    return np.array([start * (step ** i) for i in range(n)])


def sched_cos(start, end, pos):
    return start + (1 + math.cos(math.pi * (1 - pos))) * (end - start) / 2


def combined_cos(pct, start, middle, end):
    """fastai ``combined_cos``: cosine start->middle over [0,pct], middle->end after."""
    def _inner(pos):
        if pos == 1.0:
            return sched_cos(middle, end, 1.0)
        if pos >= pct:
            return sched_cos(middle, end, (pos - pct) / (1 - pct))
        return sched_cos(start, middle, pos / pct)
    return _inner


def one_cycle_scheds(lr_max: np.ndarray, moms=(0.95, 0.85, 0.95), div=25.0, div_final=1e5, pct_start=0.25):
    """``fit_one_cycle`` schedules (lr per group, momentum) as functions of pct_train."""
    lr_max = np.asarray(lr_max, dtype=np.float64)
    lr_f = combined_cos(pct_start, lr_max / div, lr_max, lr_max / div_final)
    mom_f = combined_cos(pct_start, *moms)
    return lr_f, mom_f


# --------------------------------------------------------------------------
# helpers shared by tests / bench
# --------------------------------------------------------------------------

def synthetic_batch(b, c, h, w, n_cls, seed=1234):
    """uint8/255 tiles and uniform integer masks."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (b, c, h, w), generator=g).float() / 255
    y = torch.randint(0, n_cls, (b, h, w), generator=g)
    return x, y


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())


def randomize_bn_and_zero_gammas(m: nn.Module, seed: int = 7):
    """Give BN layers non-trivial affine/running statistics so that eval-mode and
    zero-init (BatchZero) paths are actually exercised by parity tests."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for l in m.modules():
            if isinstance(l, nn.BatchNorm2d):
                l.weight.copy_(torch.rand(l.weight.shape, generator=g) * 0.5 + 0.75)
                l.bias.copy_(torch.randn(l.bias.shape, generator=g) * 0.1)
                l.running_mean.copy_(torch.randn(l.running_mean.shape, generator=g) * 0.1)
                l.running_var.copy_(torch.rand(l.running_var.shape, generator=g) * 0.5 + 0.75)
