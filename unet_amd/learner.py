"""The slice of the fastai ``Learner`` surface that the reference's scripts touch,
driving the MI355X hot path.  fastai itself is not required.

Kept: ctor kwargs ``dls, model, loss_func, opt_func, lr, splitter, cbs, metrics, path, model_dir, wd, wd_bn_bias,
train_bn, moms`` (``train.py:148-154``); ``unfreeze`` (``:246``); ``fit_one_cycle(n, lr_max=slice)`` (``:247``);
``recorder.plot_loss`` (``:253``); ``path`` / ``csv_logger.fname`` (``:257``); ``remove_cb`` / ``add_cb`` / ``dls=`` /
``loss_func=`` / ``opt_func=`` (``:226-229,258``); ``export`` (``:373``); ``load_learner`` (``:225``, ``predict.py:161``);
``predict(item)`` -> 3-tuple whose [2] is per-class probabilities [C,H,W] (``predict.py:193-203``);
``get_preds``; ``summary``; callbacks ``SaveModelCallback(monitor, comp, fname)`` and ``CSVLogger`` (``train.py:209``);
``CrossEntropyLossFlat(axis=1, weight)`` with assignable ``.func.weight`` (``train.py:195,211``), ``FocalLossFlat(gamma, axis=1)``
(``params_and_main.py:87-89``); ``DiceMulti``.
"""
from __future__ import annotations

import csv
import math
import time
from pathlib import Path
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .model import HipDynamicUnet
from .optimizer import FlatAdam, xresnet_split
from .trainer import TrainStep


# --------------------------------------------------------------------------- schedules (callback/schedule.py)

def even_mults(start: float, stop: float, n: int) -> np.ndarray:
    if n == 1:
        return np.array([stop])
    step = (stop / start) ** (1 / (n - 1))
    return np.array([start * step ** i for i in range(n)])


def _cos(a, b, pos):
    return a + (1 + math.cos(math.pi * (1 - pos))) * (b - a) / 2


def combined_cos(pct, start, middle, end):
    def f(pos):
        if pos >= 1.0:
            return _cos(middle, end, 1.0)
        if pos >= pct:
            return _cos(middle, end, (pos - pct) / (1 - pct))
        return _cos(start, middle, pos / pct)
    return f


# --------------------------------------------------------------------------- loss / metric

class _Func:
    """stand-in for ``loss_func.func`` (an nn.CrossEntropyLoss in fastai): only ``.weight`` is used by the reference"""
    def __init__(self, weight=None):
        self.weight = weight


class CrossEntropyLossFlat:
    def __init__(self, axis: int = 1, weight: Optional[torch.Tensor] = None):
        assert axis == 1
        self.axis = axis
        self.func = _Func(weight)

    def _w(self, device):
        w = self.func.weight
        return None if w is None else torch.as_tensor(w, dtype=torch.float32, device=device).contiguous()

    def __call__(self, logits: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        """Generic path (torch autograd): per-pixel weighted CE, mean = sum w[y] nll / sum w[y]."""
        return torch.nn.functional.cross_entropy(logits, targ.long(), weight=self._w(logits.device))

    def activation(self, x):
        return torch.softmax(x, dim=self.axis)

    def decodes(self, x):
        return x.argmax(dim=self.axis)


class FocalLossFlat(CrossEntropyLossFlat):
    """fastai ``FocalLossFlat(gamma=2.0, axis=1)``, the alternative classification loss of the reference's configuration
    (params_and_main.py:87-89): mean over all pixels of ``(1 - exp(-ce)) ** gamma * ce`` with ``ce = w[y] * nll`` (train.py:211 assigns
    ``.func.weight`` for every loss).  Fused on the device (unet_focal_fwd / unet_focal_bwd) like the cross-entropy."""

    def __init__(self, *args, gamma: float = 2.0, axis: int = 1, weight: Optional[torch.Tensor] = None):
        super().__init__(axis=axis, weight=weight)
        self.func.gamma = float(gamma)

    @property
    def gamma(self) -> float:
        return float(self.func.gamma)

    def __call__(self, logits: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        """Generic path (torch autograd)."""
        ce = torch.nn.functional.cross_entropy(logits, targ.long(), weight=self._w(logits.device), reduction="none")
        return ((1 - torch.exp(-ce)) ** self.gamma * ce).mean()


class _RegLoss:
    """fastai ``BaseLoss(nn.<X>Loss, axis=1, floatify=True, is_2d=False)``: prediction [B,1,H,W] and target [B,H,W] are flattened,
    'mean' reduction.  ``kind`` selects the HIP kernel of the fused training step (unet_regloss_fwd / _bwd)."""
    kind, beta = "mse", 0.5

    def __init__(self, axis: int = 1, floatify: bool = True):
        self.axis = axis
        self.func = _Func(None)       # train.py:211 assigns loss_func.func.weight for every loss; regression losses ignore it

    def __call__(self, pred: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
        """Generic path (torch autograd)."""
        d = pred.reshape(-1) - targ.float().reshape(-1)
        if self.kind == "mse":
            return (d * d).mean()
        if self.kind == "l1":
            return d.abs().mean()
        return torch.nn.functional.smooth_l1_loss(pred.reshape(-1), targ.float().reshape(-1), beta=self.beta)

    def activation(self, x):
        return x

    def decodes(self, x):
        return x


class MSELossFlat(_RegLoss):
    """train.py:191 ``MSELossFlat(axis=1)``"""
    kind = "mse"


class L1LossFlat(_RegLoss):
    kind = "l1"


class Smoothl1(_RegLoss):
    """utils.py:145-147: ``BaseLoss(nn.SmoothL1Loss, axis=1, floatify=True, is_2d=False, beta=0.5)``"""
    kind = "smoothl1"


class Rmse:
    """fastai ``rmse`` = AccumMetric(sqrt(mse), flatten=True): root of the mean squared error over the WHOLE validation set."""
    name = "_rmse"

    def reset(self):
        self.se, self.n = 0.0, 0

    def accumulate_values(self, pred: torch.Tensor, targ: torch.Tensor):
        d = (pred.reshape(-1).double() - targ.reshape(-1).double())
        self.se = self.se + (d * d).sum()          # stays a device scalar: no host sync per validation batch
        self.n += d.numel()

    # counters as a flat list of floats: Learner.validate packs every metric's counters and the loss sums into ONE all-reduce that every
    # rank takes part in, whether its validation shard held tiles or not (a shard is empty when len(valid) < world)
    def state(self, n_cls: int) -> list:
        return [float(self.se), float(self.n)]

    def load_state(self, v: list):
        self.se, self.n = float(v[0]), int(round(v[1]))

    @property
    def value(self) -> float:
        return float(np.sqrt(float(self.se) / max(1, self.n)))


class R2Score:
    """fastai ``R2Score()`` = sklearn ``r2_score`` over the whole validation set: 1 - SS_res / SS_tot."""
    name = "r2_score"

    def reset(self):
        self.n, self.st, self.stt, self.res = 0, 0.0, 0.0, 0.0

    def accumulate_values(self, pred: torch.Tensor, targ: torch.Tensor):
        p, t = pred.reshape(-1).double(), targ.reshape(-1).double()
        self.n += t.numel()
        self.st = self.st + t.sum()                # device scalars: no host sync per validation batch
        self.stt = self.stt + (t * t).sum()
        self.res = self.res + ((t - p) ** 2).sum()

    def state(self, n_cls: int) -> list:
        return [float(self.n), float(self.st), float(self.stt), float(self.res)]

    def load_state(self, v: list):
        self.n, self.st, self.stt, self.res = int(round(v[0])), float(v[1]), float(v[2]), float(v[3])

    @property
    def value(self) -> float:
        st, stt, res = float(self.st), float(self.stt), float(self.res)
        tot = stt - st * st / max(1, self.n)
        return float("nan") if tot <= 0 else float(1.0 - res / tot)


class DiceMulti:
    """fastai ``DiceMulti(axis=1)``: inter/union per class accumulated over the validation set; nanmean of 2I/U.  Device batches are counted
    by ``unet_dice_counts`` (integer histogram kernel, no host sync per batch); host tensors by ``bincount``."""
    name = "dice_multi"

    def __init__(self, axis=1):
        self.axis = axis
        self.reset()

    def reset(self):
        self._inter = self._union = None        # host-side / loaded counters (float64)
        self._counts = None                     # device counters int64 [3, C]: intersection, predicted, target

    def accumulate_argmax(self, pred: torch.Tensor, targ: torch.Tensor, n_cls: int):
        if pred.is_cuda:
            if self._counts is None:
                self._counts = torch.zeros((3, n_cls), dtype=torch.int64, device=pred.device)
            ops.dice_counts(pred.reshape(-1).long().contiguous(), targ.reshape(-1).long().contiguous(), n_cls, self._counts)
            return
        p, t = pred.reshape(-1).long(), targ.reshape(-1).long()
        cp = torch.bincount(p, minlength=n_cls)[:n_cls]
        ct = torch.bincount(t.clamp(0, n_cls - 1), minlength=n_cls)[:n_cls]
        inter = torch.bincount(p[p == t], minlength=n_cls)[:n_cls]
        self._inter = inter.double() if self._inter is None else self._inter + inter.double()
        self._union = (cp + ct).double() if self._union is None else self._union + (cp + ct).double()

    @property
    def inter(self):
        if self._counts is None:
            return self._inter
        d = self._counts[0].double()
        return d if self._inter is None else d + self._inter.to(d.device)

    @property
    def union(self):
        if self._counts is None:
            return self._union
        d = (self._counts[1] + self._counts[2]).double()
        return d if self._union is None else d + self._union.to(d.device)

    def state(self, n_cls: int) -> list:
        if self.inter is None:          # this rank saw no validation tile: it contributes zeros
            return [0.0] * (2 * n_cls)
        return self.inter.cpu().tolist() + self.union.cpu().tolist()

    def load_state(self, v: list):
        n = len(v) // 2
        self._counts = None
        self._inter, self._union = torch.tensor(v[:n], dtype=torch.float64), torch.tensor(v[n:], dtype=torch.float64)

    def all_reduce(self, device=None):
        """sum of the counters over the ranks; every rank must call it (a rank without tiles passes its class count through `reset`-time
        zeros via Learner.validate -- standalone use needs counters on every rank)"""
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.inter is None:
                raise RuntimeError("DiceMulti.all_reduce on a rank without counters: use Learner.validate, which packs zeros for it")
            i, u = self.inter.clone(), self.union.clone()
            dist.all_reduce(i); dist.all_reduce(u)
            self._counts, self._inter, self._union = None, i, u

    @property
    def value(self) -> float:
        if self.inter is None:
            return float("nan")
        i, u = self.inter.cpu().numpy(), self.union.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            s = np.where(u > 0, 2.0 * i / u, np.nan)
        return float(np.nanmean(s))


# --------------------------------------------------------------------------- data

def open_tile(path) -> np.ndarray:
    """[C,H,W] integer/float array from .npy or (uncompressed) .tif (data.py:18-28 `open_npy` reads rasters as int32)."""
    path = Path(path)
    if path.suffix == ".npy":
        a = np.load(path)
    else:
        from .tiffio import read_tiff
        a, _ = read_tiff(path)
    return a[None] if a.ndim == 2 else a


def scale_input(a: np.ndarray, dtype: str = "int8") -> np.ndarray:
    """utils.py:248-249,288-289 + IntToFloatTensor: int16 data is divided by 255 twice, int8 once."""
    x = a.astype(np.int32).astype(np.float32)     # data.py:24: cast through int32 (truncates float rasters, quirk Q6)
    if dtype == "int16":
        x = x / 255.0
    return x / 255.0


class TileDataset:
    def __init__(self, imgs: Sequence, masks: Optional[Sequence] = None, dtype: str = "int8", regression: bool = False):
        self.imgs, self.masks, self.dtype = list(imgs), None if masks is None else list(masks), dtype
        self.regression = regression      # RegressionBlock (data.py:98-99): the mask tile is a float target

    def __len__(self):
        return len(self.imgs)

    def raw(self, i) -> tuple:
        """(image [C,H,W],) or (image, mask [H,W]) as the SAMPLES the files hold: what the device feed stages (unet_amd/feed.py); scaling
        and the int64 widening of ``__getitem__`` happen on the device"""
        im = self.imgs[i]
        a = open_tile(im) if not isinstance(im, np.ndarray) else (im[None] if im.ndim == 2 else im)
        if self.masks is None:
            return (a,)
        mk = self.masks[i]
        y = np.asarray(open_tile(mk)[0] if not isinstance(mk, np.ndarray) else mk)
        if self.regression and y.dtype.kind == "f" and y.dtype != np.float32:
            y = y.astype(np.float32)
        return a, y

    def __getitem__(self, i):
        im = self.imgs[i]
        x = scale_input(open_tile(im) if not isinstance(im, np.ndarray) else im, self.dtype)
        if self.masks is None:
            return torch.from_numpy(x), None
        mk = self.masks[i]
        y = open_tile(mk)[0] if not isinstance(mk, np.ndarray) else mk
        return torch.from_numpy(x), torch.from_numpy(np.asarray(y).astype(np.float32 if self.regression else np.int64))


class DataLoader:
    """rank / world (tile-DDP, one process per GPU): every rank draws the SAME permutation (same seed) and keeps items
    rank, rank + world, ... of it.  A shuffled (training) loader truncates to a multiple of world so that all ranks run the
    same number of steps (the gradient all-reduce is a lock-step collective); a validation loader keeps every item.

    feed: "device" (default on a GPU) = unet_amd/feed.py: a decode pool fills pinned INTEGER staging buffers ``depth`` batches ahead, the
    batch is uploaded asynchronously (uint8 / uint16 samples and masks) and scaled / widened / flipped on the device; "host" = the
    reference's order of work (train.py:345, data.py:18-28, utils.py:239-295): every item opened, scaled to fp32 and stacked on the calling
    thread, then one blocking copy.  Both yield the same (xb fp32 [B,C,H,W], yb int64 [B,H,W]) bit for bit."""

    def __init__(self, ds: TileDataset, bs: int, shuffle: bool, device, drop_last: bool = False, seed: int = 0, batch_tfm=None,
                 rank: int = 0, world: int = 1, feed: str = "auto", workers: Optional[int] = None, depth: int = 3):
        self.ds, self.bs, self.shuffle, self.device, self.drop_last = ds, bs, shuffle, torch.device(device), drop_last
        self._g = np.random.default_rng(seed)
        self.batch_tfm = batch_tfm
        self.rank, self.world = rank, world
        if feed not in ("auto", "device", "host"):
            raise ValueError(f"feed = {feed!r}: 'auto', 'device' or 'host'")
        self.feed, self.workers, self.depth = feed, workers, depth
        self._feeder = None

    def shard(self, rank: int, world: int):
        self.rank, self.world = rank, world
        return self

    def _n_local(self) -> int:
        n = len(self.ds)
        if self.world == 1:
            return n
        return n // self.world if self.shuffle else len(range(self.rank, n, self.world))

    def __len__(self):
        n = self._n_local()
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _batches(self) -> list:
        idx = np.arange(len(self.ds))
        if self.shuffle:
            self._g.shuffle(idx)
        if self.world > 1:
            if self.shuffle:
                idx = idx[:len(idx) // self.world * self.world]
            idx = idx[self.rank::self.world]
        return [idx[b * self.bs:(b + 1) * self.bs] for b in range(len(self))]

    def __iter__(self):
        batches = self._batches()
        # "auto": the device feed where there is something to run ahead of -- a loader of ONE batch (Learner.predict on a single tile,
        # predict.py:193; a tiny validation set) takes the direct path instead of spinning up a thread pool and a pinned staging ring
        if self.feed == "device" or (self.feed == "auto" and self.device.type == "cuda" and len(batches) > 1):
            yield from self._iter_device(batches)
            return
        if self.feed == "auto" and self.device.type == "cuda":
            yield from self._iter_direct(batches)
            return
        for items in batches:
            items = [self.ds[int(i)] for i in items]
            xb = torch.stack([x for x, _ in items]).to(self.device)
            yb = None if items[0][1] is None else torch.stack([y for _, y in items]).to(self.device)
            if self.batch_tfm is not None and yb is not None:
                xb, yb = self.batch_tfm(xb, yb)
            yield xb, yb

    def _iter_direct(self, batches):
        """one batch, nothing to run ahead of: the items' INTEGER samples go up in one copy and are scaled / widened / flipped by the same
        kernels as the device feed (same bits as the host path), without threads or a staging ring"""
        from .feed import _SAMPLE_TYPES, as_samples
        has_y = self.ds.masks is not None
        tfm = self.batch_tfm if has_y else None

        def up(arrs):
            a = np.ascontiguousarray(np.stack([as_samples(v) for v in arrs]))      # (tiles read as band-interleaved views stack into one C-ordered block)
            t = torch.from_numpy(a.view(np.int16)).view(torch.uint16) if a.dtype == np.uint16 else torch.from_numpy(a)
            assert t.dtype == _SAMPLE_TYPES[a.dtype]
            return t.to(self.device)
        with torch.cuda.device(self.device):
            for items in batches:
                raws = [self.ds.raw(int(i)) for i in items]
                n = len(raws)
                flips = tfm.flip_flags(n) if hasattr(tfm, "flip_flags") else None
                src = up([r[0] for r in raws])
                xb = torch.empty(src.shape, dtype=torch.float32, device=self.device)
                ops.tiles_stage(src, self.ds.dtype == "int16", xb, flips)
                yb = None
                if has_y:
                    msk = up([r[1] for r in raws])
                    yb = torch.empty(msk.shape, dtype=torch.float32 if self.ds.regression else torch.int64, device=self.device)
                    ops.mask_stage(msk, yb, flips)
                if tfm is not None and flips is None:
                    xb, yb = tfm(xb, yb)
                yield xb, yb

    def _iter_device(self, batches):
        from .feed import BatchFeeder
        if self.device.type != "cuda":
            raise RuntimeError("feed='device' stages batches for the HIP kernels: it needs a GPU loader (device='cuda')")
        if self._feeder is None or self._feeder.load.__self__ is not self.ds:
            self._feeder = BatchFeeder(self.ds.raw, self.bs, self.device, depth=self.depth, workers=self.workers)
        has_y = self.ds.masks is not None
        div2 = self.ds.dtype == "int16"
        tfm = self.batch_tfm if has_y else None
        with torch.cuda.device(self.device):
            for slot in self._feeder.run(batches):
                n = slot.n
                flips = tfm.flip_flags(n) if hasattr(tfm, "flip_flags") else None       # flips run inside the staging kernels
                src = slot.dev[0][:n]
                xb = torch.empty(src.shape, dtype=torch.float32, device=self.device)
                ops.tiles_stage(src, div2, xb, flips)
                yb = None
                if has_y:
                    yb = torch.empty(slot.dev[1][:n].shape, dtype=torch.float32 if self.ds.regression else torch.int64, device=self.device)
                    ops.mask_stage(slot.dev[1][:n], yb, flips)
                slot.release()
                if tfm is not None and flips is None:
                    xb, yb = tfm(xb, yb)
                yield xb, yb


class DataLoaders:
    """``dls.train`` / ``dls.valid`` / ``dls.vocab`` / ``dls.device`` / ``dls.train_ds`` as the reference uses them."""

    def __init__(self, train: TileDataset, valid: Optional[TileDataset], bs: int, device="cuda", vocab=None, seed=0, train_tfm=None,
                 feed: str = "auto", workers: Optional[int] = None):
        self.device = torch.device(device)
        self.train_ds, self.valid_ds, self.bs, self.vocab = train, valid, bs, vocab
        self.train = DataLoader(train, bs, True, self.device, drop_last=len(train) >= bs, seed=seed, batch_tfm=train_tfm, feed=feed,
                                workers=workers)
        self.valid = None if valid is None else DataLoader(valid, bs, False, self.device, feed=feed, workers=workers)

    def test_dl(self, items, dtype=None):
        return DataLoader(TileDataset(items, None, dtype or self.train_ds.dtype), self.bs, False, self.device)


class FlipAugment:
    """The reference's default augmentation (params_and_main.py:105-115: horizontal / vertical flips) applied on the device
    with the slicing rule of SegmentationAlbumentationsTransform.encodes (utils.py:239-295): only the FIRST
    ``n_transform - B`` images of a batch are touched, n_transform = ceil(B * n_transform_imgs) -- so with the shipped default
    n_transform_imgs = 1 the slice is empty and NOTHING is augmented (quirk Q7, reproduced, not fixed)."""

    def __init__(self, p_h=0.5, p_v=0.5, n_transform_imgs=1.0, seed=0):
        self.p_h, self.p_v, self.n, self.g = p_h, p_v, n_transform_imgs, np.random.default_rng(seed)

    def __call__(self, xb: torch.Tensor, yb: torch.Tensor):
        for i, (h, v) in enumerate(self.flip_flags(xb.shape[0])):
            if h:
                xb[i] = xb[i].flip(-1); yb[i] = yb[i].flip(-1)
            if v:
                xb[i] = xb[i].flip(-2); yb[i] = yb[i].flip(-2)
        return xb, yb

    def flip_flags(self, B: int) -> list:
        """(mirror along the width, mirror along the height) per image of a batch of B: the draws of ``__call__`` in its order.  The device
        feed hands them to the staging kernels, which then write the flipped batch directly."""
        n_transform = math.ceil(B * self.n)
        flags = [(False, False)] * B
        for i in list(range(B))[:n_transform - B]:
            h = self.g.random() < self.p_h
            v = self.g.random() < self.p_v
            flags[i] = (bool(h), bool(v))
        return flags


# --------------------------------------------------------------------------- callbacks

class Callback:
    def before_fit(self, learn): ...
    def after_epoch(self, learn): ...
    def after_fit(self, learn): ...


class CSVLogger(Callback):
    def __init__(self, fname="history.csv", append=False):
        self.fname, self.append = Path(fname), append

    def before_fit(self, learn):
        self.path = learn.path / self.fname
        self.file = None
        if learn.rank != 0:          # tile-DDP: one history file, written by rank 0
            return
        self.path.parent.mkdir(parents=True, exist_ok=True)
        self.file = open(self.path, "a" if self.append else "w", newline="")
        self.writer = csv.writer(self.file)
        self.writer.writerow(learn.recorder.metric_names)

    def after_epoch(self, learn):
        if self.file is None:
            return
        self.writer.writerow(learn.recorder.log_row)
        self.file.flush()

    def after_fit(self, learn):
        if self.file is not None:
            self.file.close()


class SaveModelCallback(Callback):
    def __init__(self, monitor="valid_loss", comp=None, fname="best-model"):
        self.monitor, self.fname = monitor, fname
        self.comp = comp if comp is not None else (np.less if "loss" in monitor else np.greater)
        self.best = None

    def before_fit(self, learn):
        self.best = None

    def after_epoch(self, learn):
        val = learn.recorder.last[self.monitor]
        if self.best is None or self.comp(val, self.best):      # (the monitored values are all-reduced: same decision on every rank)
            self.best = val
            learn.save(self.fname)
            if learn.rank == 0:
                print(f"Better model found at epoch {learn.epoch} with {self.monitor} value: {val}.")

    def after_fit(self, learn):
        if self.best is not None:
            learn._barrier()
            learn.load(self.fname)


class Recorder:
    def __init__(self, metric_names):
        self.metric_names = ["epoch", "train_loss", "valid_loss"] + list(metric_names) + ["time"]
        self.losses: List[float] = []       # smoothed training loss per batch (AvgSmoothLoss beta=0.98)
        self.lrs: List[float] = []
        self.values: List[list] = []
        self.log_row: list = []
        self.last: dict = {}
        self._val, self._cnt = 0.0, 0

    def add_batch(self, loss: float, lr: float):
        self._cnt += 1
        self._val = 0.98 * self._val + 0.02 * loss          # torch.lerp(loss, val, beta)
        self.losses.append(self._val / (1 - 0.98 ** self._cnt))
        self.lrs.append(lr)

    def plot_loss(self, skip_start=5, with_valid=True):
        try:
            import matplotlib.pyplot as plt
        except Exception:
            return None
        plt.plot(list(range(skip_start, len(self.losses))), self.losses[skip_start:], label="train")
        plt.legend()
        return plt.gca()


def Adam(model, lr, mom=0.9, sqr_mom=0.99, eps=1e-5, wd=0.01, wd_bn_bias=False, splitter=xresnet_split):
    """opt_func: fastai Adam over the flat buffer (train.py:218 passes ``opt_func=Adam``)."""
    return FlatAdam(model, lr, mom, sqr_mom, eps, wd, wd_bn_bias, splitter)


# --------------------------------------------------------------------------- lr_find suggestions (fastai callback/schedule.py)

def lr_valley(lrs, losses, num_it):
    """longest strictly decreasing sub-sequence of the loss curve; suggestion = 2/3 into it"""
    n = len(losses)
    max_start = max_end = 0
    lds = [1] * n
    for i in range(1, n):
        for j in range(0, i):
            if losses[i] < losses[j] and lds[i] < lds[j] + 1:
                lds[i] = lds[j] + 1
            if lds[max_end] < lds[i]:
                max_end = i
                max_start = max_end - lds[max_end]
    sections = (max_end - max_start) / 3
    idx = max_start + int(sections) + int(sections / 2)
    return lrs[idx]


def lr_slide(lrs, losses, num_it, lr_diff=15, thresh=.005, adjust_value=1.):
    g = np.gradient(losses)
    r_idx = -1
    l_idx = r_idx - lr_diff
    local_min_lr = lrs[l_idx] if -l_idx <= len(lrs) else lrs[0]
    while l_idx >= -len(losses) and abs(g[r_idx] - g[l_idx]) > thresh:
        local_min_lr = lrs[l_idx]
        r_idx -= 1
        l_idx -= 1
    return float(local_min_lr) * adjust_value


def lr_minimum(lrs, losses, num_it):
    return lrs[int(np.argmin(losses))] / 10


def lr_steep(lrs, losses, num_it):
    grads = (losses[1:] - losses[:-1]) / (np.log(lrs[1:]) - np.log(lrs[:-1]))
    return lrs[int(np.argmin(grads))]


# --------------------------------------------------------------------------- Learner

class Learner:
    def __init__(self, dls: DataLoaders, model: HipDynamicUnet, loss_func=None, opt_func: Callable = Adam, lr=1e-3, splitter=None,
                 cbs=None, metrics=None, path=None, model_dir="models", wd=None, wd_bn_bias=False, train_bn=True,
                 moms=(0.95, 0.85, 0.95)):
        self._dls, self.model = None, model
        self.loss_func = loss_func if loss_func is not None else CrossEntropyLossFlat(axis=1)
        self.opt_func, self.lr, self.splitter = opt_func, lr, splitter or xresnet_split
        self.cbs: List[Callback] = list(cbs or [])
        self.metrics = list(metrics or [])
        self.path = Path(path) if path is not None else Path(".")
        self.model_dir, self.wd, self.wd_bn_bias, self.train_bn, self.moms = model_dir, wd, wd_bn_bias, train_bn, moms
        self.opt: Optional[FlatAdam] = None
        self.recorder = Recorder([getattr(m, "name", type(m).__name__.lower()) for m in self.metrics])
        self.epoch = 0
        # tile-DDP (one process per GPU, torch.distributed initialised by the launcher): the training loader is sharded by rank,
        # replicas start from rank 0's parameters, gradients / loss sums / metric counters are all-reduced, files are written
        # by rank 0 only.  Without an initialised process group this is the plain single-GPU Learner.
        self.rank, self.world = 0, 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dls = dls                  # through the setter: loaders are sharded by rank wherever they come from
        self._synced = False

    @property
    def dls(self):
        return self._dls

    @dls.setter
    def dls(self, dls):
        """every path that hands the Learner its loaders goes through here -- the constructor and `learn.dls = dls` of the fine-tune
        branch (reference train.py:225-229) alike -- so a tile-DDP run always trains / validates on rank shards"""
        self._dls = dls
        if dls is not None and self.world > 1:
            for dl in (getattr(dls, "train", None), getattr(dls, "valid", None)):
                if dl is not None:
                    dl.shard(self.rank, self.world)

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def _sync_replicas(self):
        """rank 0's parameters and BatchNorm buffers on every replica (model init is not seeded per rank)"""
        if self.world > 1 and not self._synced:
            from .distributed import broadcast_parameters
            broadcast_parameters(self.model.flat_param, list(self.model.buffers()))
            self.model.mark_weights_dirty()
        self._synced = True

    # -- callback plumbing the reference uses
    @property
    def csv_logger(self):
        return next(c for c in self.cbs if isinstance(c, CSVLogger))

    def add_cb(self, cb):
        self.cbs.append(cb)
        return self

    def remove_cb(self, cb):
        self.cbs = [c for c in self.cbs if not (c is cb or (isinstance(cb, type) and isinstance(c, cb)))]
        return self

    def unfreeze(self):
        return self          # every parameter group trains (train_bn=True, nothing is frozen on this path)

    def create_opt(self):
        kw = {} if self.wd is None else {"wd": self.wd}
        self.opt = self.opt_func(self.model, self.lr, wd_bn_bias=self.wd_bn_bias, splitter=self.splitter, **kw)

    def _weights(self):
        return self.loss_func._w(self.dls.device) if isinstance(self.loss_func, CrossEntropyLossFlat) else None

    @property
    def _focal_gamma(self) -> Optional[float]:
        return self.loss_func.gamma if isinstance(self.loss_func, FocalLossFlat) else None

    @property
    def regression(self) -> bool:
        return isinstance(self.loss_func, _RegLoss)

    # -- training
    def fit_one_cycle(self, n_epoch, lr_max=None, div=25.0, div_final=1e5, pct_start=0.25, wd=None, moms=None):
        if self.opt is None:
            self.create_opt()
        if wd is not None:
            self.opt.wd = wd
        k = len(self.opt.groups)
        if lr_max is None:
            lr_max = self.lr
        if isinstance(lr_max, slice):
            lrs = even_mults(lr_max.start, lr_max.stop, k) if lr_max.start else np.array([lr_max.stop / 10] * (k - 1) + [lr_max.stop])
        else:
            lrs = np.array([float(lr_max)] * k)
        lr_f = combined_cos(pct_start, lrs / div, lrs, lrs / div_final)
        mom_f = combined_cos(pct_start, *(self.moms if moms is None else moms))
        self._fit(n_epoch, lr_f, mom_f)

    def fit(self, n_epoch, lr=None):
        if self.opt is None:
            self.create_opt()
        lrs = np.array([self.lr if lr is None else lr] * len(self.opt.groups), dtype=np.float64)
        self._fit(n_epoch, lambda p: lrs, lambda p: self.opt.mom)

    def _fit(self, n_epoch, lr_f, mom_f, batch_cb=None):
        """batch_cb(it, loss_tensor, lr) -> True stops the fit after that batch (lr_find)."""
        model, opt = self.model, self.opt
        self._sync_replicas()
        fused = isinstance(self.loss_func, (CrossEntropyLossFlat, _RegLoss))
        if self.world > 1 and not fused:
            raise RuntimeError("tile-DDP needs one of the fused losses (CrossEntropyLossFlat / MSELossFlat / L1LossFlat / Smoothl1)")
        step = TrainStep(model, opt, self._weights(), self.world) if fused else None
        if fused and self.regression:
            step.reg_kind, step.reg_beta = self.loss_func.kind, self.loss_func.beta
        if fused:
            step.focal_gamma = self._focal_gamma
        n_iter = len(self.dls.train)
        total = max(1, n_epoch * n_iter)
        for cb in self.cbs:
            cb.before_fit(self)
        it = 0
        for self.epoch in range(n_epoch):
            t0 = time.time()
            model.train()
            pending = []
            for xb, yb in self.dls.train:
                pct = it / total
                opt.set_lr(lr_f(pct)); opt.mom = float(mom_f(pct))
                if fused:
                    step.weights = self._weights()
                    loss = step(xb, yb)
                else:
                    loss = self.loss_func(model(xb), yb)
                    model.flat_grad.zero_()
                    loss.backward()
                    opt.step()
                pending.append((loss.detach().reshape(1).clone(), opt.lrs[-1]))
                it += 1
                if batch_cb is not None and batch_cb(it, pending[-1][0], opt.lrs[-1]):
                    return
            for l, lr in pending:                       # one host sync per epoch, not per batch
                self.recorder.add_batch(float(l.item()), lr)
            train_loss = self.recorder.losses[-1] if self.recorder.losses else float("nan")
            vals = self.validate()
            el = int(time.time() - t0)
            row = [self.epoch, train_loss] + vals + [f"{el // 60:02d}:{el % 60:02d}"]
            self.recorder.values.append(row[1:-1])
            self.recorder.log_row = row
            self.recorder.last = dict(zip(self.recorder.metric_names[1:-1], row[1:-1]))
            if not getattr(self, "_no_logging", False) and self.rank == 0:
                print(dict(zip(self.recorder.metric_names, row)))
            for cb in self.cbs:
                cb.after_epoch(self)
        for cb in self.cbs:
            cb.after_fit(self)

    @torch.no_grad()
    def validate(self, dl=None) -> list:
        dl = dl or self.dls.valid
        if dl is None:
            return [float("nan")] + [float("nan")] * len(self.metrics)
        model = self.model
        model.eval()
        for m in self.metrics:
            m.reset()
        ctx = model.ctx
        w = self._weights()
        acc = torch.zeros(2, dtype=torch.float64, device=model._device)      # loss numerator / denominator: summed on the device, read ONCE
        for xb, yb in dl:
            z = model._hip_forward(xb.to(model._device, torch.float32), False)
            if self.regression:
                yb = yb.to(model._device, torch.float32).contiguous()
                loss = ctx.vec(self, "vloss", 1)
                ops.regloss_fwd(z, yb, self.loss_func.kind, self.loss_func.beta, loss, ctx.workspace(ops.ce_workspace(z.P)))
                vals = torch.empty((z.N, z.C, z.H, z.W), dtype=torch.float32, device=model._device)
                ops.nhwc_to_nchw(z, vals)
                acc[0] += loss[0].double() * z.P
                acc[1] += z.P
                for m in self.metrics:
                    m.accumulate_values(vals[:, 0], yb)
                continue
            yb = yb.to(model._device, torch.int64).contiguous()
            loss, denom = ctx.vec(self, "vloss", 1), ctx.vec(self, "vden", 1)
            if self._focal_gamma is not None:          # a plain mean over the pixels: numerator = loss * P, denominator = P
                ops.focal_fwd(z, yb, w, self._focal_gamma, loss, ctx.workspace(ops.ce_workspace(z.P)))
                acc[0] += loss[0].double() * z.P
                acc[1] += z.P
            else:
                ops.ce_fwd(z, yb, w, loss, denom, ctx.workspace(ops.ce_workspace(z.P)))
                acc[0] += loss[0].double() * denom[0].double()
                acc[1] += denom[0].double()
            amax = torch.empty((z.N, z.H, z.W), dtype=torch.int64, device=model._device)
            ops.softmax_argmax(z, None, amax)
            for m in self.metrics:
                m.accumulate_argmax(amax, yb, z.C)
        num, den = acc.cpu().tolist()          # the one host sync of the pass
        if self.world > 1:
            # ONE collective per validation pass, the same on every rank whether or not its shard held a tile: valid_loss = sum of the
            # ranks' numerators / sum of their denominators, followed by every metric's counters
            import torch.distributed as dist
            states = [m.state(model.n_out) for m in self.metrics]
            t = torch.tensor([num, den] + [v for st in states for v in st], dtype=torch.float64, device=model._device)
            dist.all_reduce(t)
            vals = t.cpu().tolist()
            num, den = vals[0], vals[1]
            o = 2
            for m, st in zip(self.metrics, states):
                m.load_state(vals[o:o + len(st)])
                o += len(st)
        return [num / max(den, 1e-30)] + [m.value for m in self.metrics]

    # -- inference (predict.py:193)
    @staticmethod
    def _to_host(parts: list) -> Optional[torch.Tensor]:
        """device batches -> ONE host tensor: concatenated on the device, copied into a pinned buffer of torch's caching host allocator
        (a fresh pageable tensor of a few MB costs several ms of page faults on a memory-capped box: 50 of the 53 ms Learner.predict took
        per tile before round 5 were such allocations and three pageable copies)"""
        if not parts:
            return None
        t = parts[0] if len(parts) == 1 else torch.cat(parts)
        if not t.is_cuda:
            return t
        out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        out.copy_(t)                      # (blocking on the stream: the tensor is complete when it is handed out)
        return out

    @torch.no_grad()
    def get_preds(self, dl=None, with_input=False, with_decoded=False):
        dl = dl or self.dls.valid
        self.model.eval()
        xs, ps, ys, ds = [], [], [], []
        for xb, yb in dl:
            if self.regression:
                vals = self.model.predict_values(xb)      # decoded == preds: no activation (train.py:90-95)
                ps.append(vals)
            else:
                probs, amax = self.model.predict_probs(xb)
                ps.append(probs); ds.append(amax)
            if with_input:
                xs.append(xb)
            if yb is not None:
                ys.append(yb)
        preds = self._to_host(ps)
        res = (preds, self._to_host(ys))
        if with_decoded:
            res = res + (preds if self.regression else self._to_host(ds),)
        if with_input:
            res = (self._to_host(xs),) + res
        return res

    def predict(self, item, rm_type_tfms=None, with_input=False):
        """(decoded mask, argmax [H,W], per-class probabilities [C,H,W]) for one tile (path or [C,H,W] array); in regression
        mode the 2-tuple (decoded, preds) of ``Learner_adjust.predict`` (train.py:87-95), both [1,H,W].  The reference's per-tile loop
        (predict.py:191-193); predict.save_predictions batches 16 tiles instead."""
        dl = self.dls.test_dl([item])
        preds, _, dec = self.get_preds(dl=dl, with_decoded=True)
        if self.regression:
            return dec[0], preds[0]
        res = dec[0], dec[0], preds[0]
        return res

    # -- learning-rate finder (fastai callback/schedule.py LRFinder + Learner.lr_find; reference utils.py:150-167)
    def lr_find(self, start_lr=1e-7, end_lr=10, num_it=100, stop_div=True, show_plot=False, suggest_funcs=("valley",)):
        """Exponential LR sweep over ``num_it`` training batches (all parameter groups at the same rate), stopped early when
        the smoothed loss exceeds 4x its best; model and optimizer state are restored afterwards.  Returns an object with one
        attribute per suggestion function (``.valley``, ``.slide``, ``.steep``, ``.minimum``)."""
        from types import SimpleNamespace
        if self.opt is None:
            self.create_opt()
        num_it = max(6, int(num_it))
        tmp = f"_tmp_lr_find_r{self.rank}"
        self._save_local(tmp, with_opt=True)
        rec, self.recorder = self.recorder, Recorder([])
        self._no_logging = True
        cbs, self.cbs = self.cbs, []
        valid, self.dls.valid = self.dls.valid, None          # before_validate: CancelValidException
        k = len(self.opt.groups)
        best = [float("inf")]
        smooth = Recorder([])

        def lr_f(_pct, it=[0]):
            lr = start_lr * (end_lr / start_lr) ** (it[0] / num_it)       # SchedExp at pos = train_iter / num_it
            it[0] += 1
            return np.array([lr] * k)

        def batch_cb(it, loss, lr):
            smooth.add_batch(float(loss.item()), lr)
            sl = smooth.losses[-1]
            best[0] = min(best[0], sl)
            return (stop_div and (sl > 4 * best[0] or not np.isfinite(sl))) or it >= num_it
        n_epoch = num_it // max(1, len(self.dls.train)) + 1
        try:
            self._fit(n_epoch, lr_f, lambda p: self.opt.mom, batch_cb=batch_cb)
        finally:
            self.recorder, self.cbs, self.dls.valid = rec, cbs, valid
            self._no_logging = False
            self.model.flat_grad.zero_()
            self.load(tmp, with_opt=True)
            self.model.mark_weights_dirty()
            try:
                self._model_path(tmp).unlink()
            except OSError:
                pass
        lrs = np.array(smooth.lrs[num_it // 10:-5], dtype=np.float64)
        losses = np.array(smooth.losses[num_it // 10:-5], dtype=np.float64)
        ok = np.isfinite(losses)
        lrs, losses = lrs[ok], losses[ok]
        self.lr_find_curve = (smooth.lrs, smooth.losses)
        fns = {"valley": lr_valley, "slide": lr_slide, "steep": lr_steep, "minimum": lr_minimum}
        out = {}
        for f in suggest_funcs:
            name = f if isinstance(f, str) else f.__name__.replace("lr_", "")
            out[name] = float(fns[name](lrs, losses, num_it)) if len(lrs) > 1 else float("nan")
        return SimpleNamespace(**out)

    # -- persistence
    def _model_path(self, name):
        p = self.path / self.model_dir
        p.mkdir(parents=True, exist_ok=True)
        return p / f"{name}.pth"

    def _save_local(self, name, with_opt=False):
        """fastai ``save_model`` file layout: the bare model ``state_dict`` (what SaveModelCallback writes: with_opt=False), or
        ``{'model': ..., 'opt': ...}`` with the optimizer state."""
        sd = {k: v.cpu() for k, v in self.model.state_dict().items()}
        if with_opt and self.opt is not None:
            sd = {"model": sd, "opt": {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.opt.state_dict().items()}}
        torch.save(sd, self._model_path(name))

    def save(self, name, with_opt=False):
        if self.rank == 0:           # tile-DDP: replicas are identical, one writer
            self._save_local(name, with_opt)
        self._barrier()

    def load(self, name, with_opt=False):
        """fastai ``load_model``: accepts both layouts; an optimizer state that is not this optimizer's (e.g. written by
        fastai itself) is skipped with a warning, as fastai does."""
        sd = torch.load(self._model_path(name), map_location="cpu")
        hasopt = isinstance(sd, dict) and set(sd.keys()) == {"model", "opt"}
        self.model.load_state_dict(sd["model"] if hasopt else sd)
        if hasopt and with_opt and self.opt is not None:
            try:
                self.opt.load_state_dict({k: (v.to(self.model._device) if torch.is_tensor(v) else v) for k, v in sd["opt"].items()})
            except Exception as e:      # noqa: BLE001
                import warnings
                warnings.warn(f"Could not load the optimizer state ({type(e).__name__}: {e}); the model weights were loaded.")
        elif with_opt and not hasopt:
            import warnings
            warnings.warn("Saved file doesn't contain an optimizer state.")
        return self

    def export(self, fname="export.pkl"):
        """state dict + the constructor arguments (fastai pickles the whole Learner; that pickle needs fastai to load)."""
        m = self.model
        w = self.loss_func.func.weight if isinstance(self.loss_func, CrossEntropyLossFlat) else None
        meta = {"arch": m.arch, "n_in": m.n_in, "n_out": m.n_out, "img_size": list(m.img_size), "vocab": self.dls.vocab,
                "dtype": self.dls.train_ds.dtype if self.dls is not None else "int8",
                "class_weights": None if w is None else [float(v) for v in torch.as_tensor(w).cpu()],
                "regression": self.loss_func.kind if self.regression else None,
                "focal_gamma": self._focal_gamma,
                "self_attention": bool(getattr(m, "self_attention", False)), "act_dtype": getattr(m, "act_dtype", "f32")}
        p = Path(fname)
        p = p if p.is_absolute() else self.path / p
        if self.rank == 0:
            p.parent.mkdir(parents=True, exist_ok=True)
            torch.save({"meta": meta, "model": {k: v.cpu() for k, v in m.state_dict().items()}}, p)
        self._barrier()

    def summary(self) -> str:
        m = self.model
        n = sum(p.numel() for p in m.parameters())
        lines = [f"HipDynamicUnet({m.arch}, n_in={m.n_in}, n_out={m.n_out}, img_size={m.img_size})", f"Total params: {n:,}",
                 f"Parameter groups: {[sum(p.numel() for p in g) for g in self.splitter(m)]}",
                 f"Loss: {type(self.loss_func).__name__}  Optimizer: fastai Adam (flat, HIP)"]
        return "\n".join(lines)


class Learner_adjust(Learner):
    """train.py:87-95: the regression Learner (predict returns the 2-tuple); the behaviour lives in Learner.predict."""


def load_learner(fname, device="cuda", act_dtype: Optional[str] = None) -> Learner:
    """learner of an exported model file.  act_dtype None = the storage mode the model was exported with ("f32" for files written before
    the mode was recorded); self-attention is rebuilt when the file says so or, for such older files, when the state dict holds its keys."""
    d = torch.load(fname, map_location="cpu")
    meta = d["meta"]
    sa = meta.get("self_attention")
    if sa is None:
        sa = any(".query." in k for k in d["model"])
    from .model import skip_weight_init
    with skip_weight_init():          # every parameter and buffer comes from the file (strict load below)
        model = HipDynamicUnet(meta["arch"], meta["n_in"], meta["n_out"], tuple(meta["img_size"]), self_attention=bool(sa), device=device,
                               act_dtype=act_dtype or meta.get("act_dtype", "f32"))
    model.load_state_dict(d["model"], strict=True)
    empty = TileDataset([], None, meta.get("dtype", "int8"))
    dls = DataLoaders(empty, None, 1, device=device, vocab=meta.get("vocab"))
    w = meta.get("class_weights")
    if meta.get("regression"):
        loss = {"mse": MSELossFlat, "l1": L1LossFlat, "smoothl1": Smoothl1}[meta["regression"]](axis=1)
        return Learner_adjust(dls, model, loss_func=loss, metrics=[Rmse(), R2Score()])
    wt = None if w is None else torch.tensor(w)
    loss = CrossEntropyLossFlat(axis=1, weight=wt) if meta.get("focal_gamma") is None else FocalLossFlat(gamma=meta["focal_gamma"], axis=1, weight=wt)
    return Learner(dls, model, loss_func=loss, metrics=[DiceMulti()])
