"""Training entry points of the tile workflow on the MI355X hot path.

Mirrors the reference's ``train.py`` surface: ``unet_learner_MS`` (``train.py:98-160``), ``train_unet`` (``:163-283``) and
``train_func`` with its 25 positional arguments (``:287-292``).  Data layout: ``<data_path>/{trai,vali}/{img_tiles,mask_tiles}``
with ``.tif`` (uncompressed GeoTIFF) or ``.npy`` tiles.  Regression mode (``enable_regression``: n_out = 1, MSELossFlat /
L1LossFlat / Smoothl1, rmse + R2Score, ``Learner_adjust``) and the learning-rate finder (``LR_FINDER`` = valley / slide /
steep / minimum, utils.py:150-167) are available.  Out of scope here: plots, albumentations pipelines
(only the built-in flips), run-parameter JSON beyond a compact dump.
"""
from __future__ import annotations

import json
import os
import shutil
import warnings
from pathlib import Path

import numpy as np
import torch

from unet_amd import xresnet34  # noqa: F401  (architecture tokens, like `from fastai.vision.all import xresnet34`)
from unet_amd.learner import (Adam, CrossEntropyLossFlat, CSVLogger, DataLoaders, DiceMulti, FlipAugment, FocalLossFlat, L1LossFlat, Learner,  # noqa: F401
                              Learner_adjust, MSELossFlat, R2Score, Rmse, SaveModelCallback, Smoothl1, TileDataset, load_learner,
                              open_tile)
from unet_amd.model import HipDynamicUnet


def _arch_name(arch) -> str:
    return arch if isinstance(arch, str) else arch.__name__


def _tiles(folder: Path):
    return sorted([p for p in folder.iterdir() if p.suffix.lower() in (".tif", ".tiff", ".npy")])


def get_datatype(data_path: Path) -> str:
    """utils.py:72-89: max < 257 in the first training tile -> 'int8' else 'int16'."""
    first = _tiles(Path(data_path) / "trai" / "img_tiles")[0]
    return "int8" if open_tile(first).max() < 257 else "int16"


def get_class_weights(ds: TileDataset, n_cls: int, max_tiles: int = 1200) -> np.ndarray:
    """utils.py:106-117: total / count per class over (up to 1200) training masks."""
    # only the MASK tiles are opened (ds[i] would also read and scale the image: 5 MB of float work per tile for nothing), by a few threads
    from concurrent.futures import ThreadPoolExecutor

    def count(i):
        mk = ds.masks[i]
        y = np.asarray(mk if isinstance(mk, np.ndarray) else open_tile(mk)[0]).astype(np.int64).ravel()
        return np.bincount(y, minlength=n_cls)[:n_cls]
    cnt = np.zeros(n_cls, dtype=np.float64)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for c in ex.map(count, range(min(len(ds), max_tiles))):
            cnt += c
    return cnt.sum() / np.maximum(cnt, 1)


def make_dataloaders(data_path, bs, codes, dtype=None, device="cuda", train_tfm=None, regression=False) -> DataLoaders:
    data_path = Path(data_path)
    dtype = dtype or get_datatype(data_path)
    sets = {}
    for split in ("trai", "vali"):
        imgs = _tiles(data_path / split / "img_tiles")
        masks = [data_path / split / "mask_tiles" / p.name for p in imgs]
        sets[split] = TileDataset(imgs, masks, dtype, regression=regression)
    return DataLoaders(sets["trai"], sets["vali"], bs, device=device, vocab=list(codes), train_tfm=train_tfm)


def unet_learner_MS(dls, arch, pretrained=True, loss_func=None, norm_type=None, opt_func=Adam, lr=1e-3, splitter=None, cbs=None,
                    metrics=None, path=None, model_dir="models", wd=None, wd_bn_bias=False, train_bn=True,
                    moms=(0.95, 0.85, 0.95), regression=False, self_attention=False) -> Learner:
    x0, _ = dls.train_ds[0]
    n_in, size = x0.shape[0], tuple(x0.shape[-2:])
    n_out = 1 if regression else len(dls.vocab)                      # train.py:137-140
    # UNET_ACT_DTYPE=bf16 selects the bf16-storage variant (BASELINE configs[1]); the reference has no such switch and computes in fp32
    model = HipDynamicUnet(_arch_name(arch), n_in, n_out, size, self_attention=self_attention, device=dls.device,
                           act_dtype=os.environ.get("UNET_ACT_DTYPE", "f32"))
    cls = Learner_adjust if regression else Learner                   # train.py:148
    return cls(dls=dls, model=model, loss_func=loss_func, opt_func=opt_func, lr=lr, splitter=splitter, cbs=cbs, metrics=metrics,
                   path=path, model_dir=model_dir, wd=wd, wd_bn_bias=wd_bn_bias, train_bn=train_bn, moms=moms)


def find_lr(learn, finder):
    """utils.py:150-167: suggested maximum learning rate from the LR finder."""
    lrs = learn.lr_find(suggest_funcs=("minimum", "steep", "valley", "slide"), show_plot=False)
    if finder not in ("valley", "slide", "steep", "minimum"):
        warnings.warn("Learning rate finder parameter not recognised (minimum, steep, valley, slide, None). Using valley.")
        finder = "valley"
    return getattr(lrs, finder)


def train_unet(class_weights, dls, architecture, epochs, path, lr, encoder_factor, lr_finder=None, regression=False,
               loss_func=None, monitor=None, existing_model=None, self_attention=False, export_model_summary=False) -> Learner:
    weights = torch.tensor(np.asarray(class_weights, dtype=np.float32), device=dls.device)
    if regression:                                                    # train.py:189-193
        if loss_func is None:
            loss_func = MSELossFlat(axis=1)
        metrics = [Rmse(), R2Score()]
    else:
        if loss_func is None:
            loss_func = CrossEntropyLossFlat(axis=1, weight=weights)
        metrics = [DiceMulti()]
    monitor = monitor or ("r2_score" if regression else "dice_multi")
    comp = np.less if monitor in ("train_loss", "valid_loss") else np.greater
    if monitor not in ("train_loss", "valid_loss", "r2_score", "dice_multi"):
        warnings.warn("Monitor not recognised. Assuming maximization.")
    path = Path(path)
    cbs = [SaveModelCallback(monitor=monitor, comp=comp, fname="best-model"), CSVLogger()]
    loss_func.func.weight = weights                     # train.py:211 (also overrides a user loss: quirk Q5)
    if existing_model is None:
        learn = unet_learner_MS(dls, architecture, loss_func=loss_func, opt_func=Adam, metrics=metrics, cbs=cbs,
                                regression=regression, self_attention=self_attention, path=path.parent)
    else:
        learn = load_learner(existing_model, device=dls.device)
        learn.dls, learn.loss_func, learn.opt_func, learn.path = dls, loss_func, Adam, path.parent
        learn.cbs = cbs
    if export_model_summary and learn.rank == 0:
        Path(str(path).rsplit(".", 1)[0] + "_model_summary.txt").write_text(
            f"Class_weights: {class_weights}\n{learn.summary()}\n{learn.model}\n")
    if lr_finder is not None:
        lr = find_lr(learn, lr_finder)
        if learn.rank == 0:
            print(f"Optimized learning rate: {lr}")
    learn.unfreeze()
    learn.fit_one_cycle(epochs, lr_max=slice(lr / encoder_factor, lr))
    hist = Path(str(path).rsplit(".", 1)[0] + "_history.csv")
    if learn.rank == 0:                                 # tile-DDP: rank 0 owns the files
        shutil.move(str(learn.path / learn.csv_logger.fname), str(hist))
    learn._barrier()
    learn.remove_cb(CSVLogger)
    return learn


def train_func(data_path, existing_model, model_Path, description, BATCH_SIZE, visualize_data_example, enable_regression,
               CLASS_WEIGHTS, ARCHITECTURE, EPOCHS, LEARNING_RATE, ENCODER_FACTOR, LR_FINDER, loss_func, monitor, self_attention,
               VALID_SCENES, CODES, transforms, split_idx, export_model_summary, aug_pipe, n_transform_imgs, info, class_zero):
    # one process per GPU under torch.distributed.run (tile-DDP: BASELINE configs[2]); a plain `python train.py ...` is world 1
    from unet_amd.distributed import init_from_env
    rank, local_rank, world = init_from_env()
    device = f"cuda:{local_rank}" if world > 1 else "cuda"
    data_path = Path(data_path)
    dtype = get_datatype(data_path)
    new_path = Path(model_Path) / description
    new_path.mkdir(parents=True, exist_ok=True)
    model_path = new_path / f"{description}.pkl"
    if rank == 0:
        (new_path / f"{description}.json").write_text(json.dumps({
            "data_path": str(data_path), "BATCH_SIZE": BATCH_SIZE, "EPOCHS": EPOCHS, "LEARNING_RATE": LEARNING_RATE,
            "ENCODER_FACTOR": ENCODER_FACTOR, "CLASS_WEIGHTS": CLASS_WEIGHTS if isinstance(CLASS_WEIGHTS, str) else list(CLASS_WEIGHTS),
            "ARCHITECTURE": _arch_name(ARCHITECTURE), "CODES": list(CODES), "self_attention": self_attention, "monitor": monitor,
            "VALID_SCENES": VALID_SCENES, "info": info, "class_zero": class_zero, "dtype": dtype, "world_size": world}, indent=1, default=str))
    tfm = None
    if transforms:
        # aug_pipe: None = the reference's default pipeline HorizontalFlip + VerticalFlip (params_and_main.py:105-115); or a
        # unet_amd.augment.Compose (HorizontalFlip / VerticalFlip / RandomBrightnessContrast / CoarseDropout with albumentations'
        # semantics, applied on the device); a ready batch transform (FlipAugment / BatchAugment) is used as is
        from unet_amd.augment import BatchAugment, Compose
        if isinstance(aug_pipe, (FlipAugment, BatchAugment)):
            tfm = aug_pipe
        elif isinstance(aug_pipe, Compose):
            tfm = BatchAugment(aug_pipe, n_transform_imgs=n_transform_imgs)
        else:
            if aug_pipe is not None:
                warnings.warn("aug_pipe is not a unet_amd.augment.Compose (albumentations itself is not available on this path); "
                              "using the default flip pipeline")
            tfm = FlipAugment(n_transform_imgs=n_transform_imgs)
    dls = make_dataloaders(data_path, BATCH_SIZE, CODES, dtype, device=device, train_tfm=tfm, regression=bool(enable_regression))
    if enable_regression:
        CLASS_WEIGHTS = [1]                                           # train.py:334-335
    elif isinstance(CLASS_WEIGHTS, str):
        if CLASS_WEIGHTS == "even":
            CLASS_WEIGHTS = np.ones(len(CODES)) / len(CODES)
        elif CLASS_WEIGHTS == "weighted":
            CLASS_WEIGHTS = get_class_weights(dls.train_ds, len(CODES))
    if rank == 0:
        print(f"Train files: {len(dls.train_ds)}, Test files: {len(dls.valid_ds)}" + (f" (sharded over {world} ranks)" if world > 1 else ""))
        print(f"Class weights: {CLASS_WEIGHTS}")
    learn = train_unet(class_weights=CLASS_WEIGHTS, dls=dls, architecture=ARCHITECTURE, epochs=EPOCHS, path=model_path,
                       lr=LEARNING_RATE, encoder_factor=ENCODER_FACTOR, lr_finder=LR_FINDER, regression=enable_regression,
                       loss_func=loss_func, monitor=monitor, existing_model=existing_model, self_attention=self_attention,
                       export_model_summary=export_model_summary)
    learn.export(model_path)
    return learn
