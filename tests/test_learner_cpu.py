"""CPU suite for the host-side Learner surface: schedules, recorder smoothing, DiceMulti, CSV format, GeoTIFF I/O,
input scaling -- checked against the oracle's restatement of fastai where one exists."""
import math

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_amd import learner as L
from unet_amd.tiffio import read_tiff, write_tiff


def test_schedules_match_oracle():
    lrs = L.even_mults(1e-4, 1e-3, 3)
    assert np.allclose(lrs, O.even_mults(1e-4, 1e-3, 3))
    f, g = L.combined_cos(0.25, lrs / 25, lrs, lrs / 1e5), O.combined_cos(0.25, lrs / 25, lrs, lrs / 1e5)
    m, n = L.combined_cos(0.25, 0.95, 0.85, 0.95), O.combined_cos(0.25, 0.95, 0.85, 0.95)
    for p in np.linspace(0, 1, 41):
        assert np.allclose(f(float(p)), g(float(p))) and abs(m(float(p)) - n(float(p))) < 1e-12


def test_dice_multi_matches_oracle():
    g = torch.Generator().manual_seed(0)
    a, b = L.DiceMulti(), O.DiceMulti()
    for _ in range(3):
        logits = torch.randn(2, 4, 9, 7, generator=g)
        targ = torch.randint(0, 4, (2, 9, 7), generator=g)
        a.accumulate_argmax(logits.argmax(1), targ, 4)
        b.accumulate(logits, targ)
    assert abs(a.value - b.value) < 1e-12
    c = L.DiceMulti()
    c.accumulate_argmax(torch.zeros(1, 2, 2, dtype=torch.long), torch.zeros(1, 2, 2, dtype=torch.long), 3)
    assert c.value == 1.0          # absent classes are skipped (nanmean)


def test_recorder_smoothing_is_fastai_avgsmoothloss():
    r = L.Recorder(["dice_multi"])
    assert r.metric_names == ["epoch", "train_loss", "valid_loss", "dice_multi", "time"]     # history.csv:1 of the reference
    val, vals = 0.0, [2.0, 1.0, 0.5]
    for i, v in enumerate(vals, 1):
        r.add_batch(v, 1e-3)
        val = 0.98 * val + 0.02 * v
        assert abs(r.losses[-1] - val / (1 - 0.98 ** i)) < 1e-12


def test_cross_entropy_flat_surface():
    w = torch.tensor([0.2, 0.3, 0.5])
    lf = L.CrossEntropyLossFlat(axis=1)
    lf.func.weight = w                                  # train.py:211 assigns it
    z = torch.randn(2, 3, 4, 5)
    y = torch.randint(0, 3, (2, 4, 5))
    assert abs(float(lf(z, y)) - float(O.CrossEntropyLossFlat(weight=w)(z, y))) < 1e-6
    assert torch.equal(lf.decodes(z), z.argmax(1)) and torch.allclose(lf.activation(z).sum(1), torch.ones(2, 4, 5))


def test_input_scaling_quirk():
    a = np.array([[[0, 255], [128, 64]]], dtype=np.uint8)
    assert np.allclose(L.scale_input(a, "int8"), a / 255.0)
    b = np.array([[[0, 65025]]], dtype=np.uint16)
    assert np.allclose(L.scale_input(b, "int16"), b / 255.0 / 255.0)      # utils.py:248-249 + IntToFloatTensor


@pytest.mark.parametrize("dt", [np.uint8, np.uint16, np.int16, np.float32])
def test_geotiff_roundtrip(tmp_path, dt):
    a = (np.random.default_rng(0).random((4, 13, 17)) * 200).astype(dt)
    gt = (500000.0, 0.2, 0.0, 5800000.0, 0.0, -0.2)
    write_tiff(tmp_path / "a.tif", a, geotransform=gt, nodata=0)
    b, meta = read_tiff(tmp_path / "a.tif")
    assert b.dtype == a.dtype and np.array_equal(a, b) and meta["geotransform"] == gt and meta["nodata"] == 0.0
    write_tiff(tmp_path / "m.tif", a[0])
    m, meta = read_tiff(tmp_path / "m.tif")
    assert m.shape == (13, 17) and np.array_equal(m, a[0]) and meta["geotransform"] is None


def test_geotiff_reads_foreign_writer(tmp_path):
    from PIL import Image
    a = (np.random.default_rng(1).random((9, 11)) * 255).astype(np.uint8)
    Image.fromarray(a).save(tmp_path / "p.tif")
    b, _ = read_tiff(tmp_path / "p.tif")
    assert np.array_equal(a, b)
    rgb = (np.random.default_rng(2).random((8, 8, 3)) * 255).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "c.tif")
    c, _ = read_tiff(tmp_path / "c.tif")
    assert np.array_equal(c, np.moveaxis(rgb, -1, 0))
    Image.fromarray(a).save(tmp_path / "z.tif", compression="tiff_lzw")          # (round 4: LZW / Deflate / PackBits strips are read)
    z, _ = read_tiff(tmp_path / "z.tif")
    assert np.array_equal(a, z)
    Image.fromarray(rgb).save(tmp_path / "j.tif", compression="jpeg")            # (round 5: JPEG-in-TIFF is read, to libtiff's own bytes)
    j, _ = read_tiff(tmp_path / "j.tif")
    assert np.array_equal(j, np.moveaxis(np.asarray(Image.open(tmp_path / "j.tif")), -1, 0))
    Image.fromarray(a > 127).save(tmp_path / "g4.tif", compression="group4")     # what the reader does not decode raises, loudly
    with pytest.raises(NotImplementedError):
        read_tiff(tmp_path / "g4.tif")


def test_dataloaders_batches(tmp_path):
    g = np.random.default_rng(3)
    imgs = [g.integers(0, 255, (4, 8, 8)).astype(np.uint8) for _ in range(5)]
    masks = [g.integers(0, 3, (8, 8)).astype(np.uint8) for _ in range(5)]
    dls = L.DataLoaders(L.TileDataset(imgs, masks), L.TileDataset(imgs[:2], masks[:2]), bs=2, device="cpu", vocab=["a", "b", "c"])
    xs = list(dls.train)
    assert len(xs) == 2 and xs[0][0].shape == (2, 4, 8, 8) and xs[0][0].dtype == torch.float32 and xs[0][1].dtype == torch.int64
    assert float(xs[0][0].max()) <= 1.0
    assert len(list(dls.valid)) == 1


# ------------------------------------------------------------------ regression mode / lr_find host logic

@pytest.mark.parametrize("cls", ["MSELossFlat", "L1LossFlat", "Smoothl1"])
def test_regression_losses_match_oracle(cls):
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(3, 1, 6, 7, generator=g)
    targ = torch.randn(3, 6, 7, generator=g) * 2
    mine, ref = getattr(L, cls)(axis=1), getattr(O, cls)(axis=1)
    mine.func.weight = torch.ones(1)                  # train.py:211 assigns it even in regression mode; it must be accepted and ignored
    assert abs(float(mine(pred, targ)) - float(ref(pred, targ))) < 1e-6
    assert mine.decodes(pred) is pred and mine.activation(pred) is pred
    assert mine.kind in ("mse", "l1", "smoothl1")


def test_rmse_r2_accumulate_over_batches():
    g = torch.Generator().manual_seed(6)
    a, b = L.Rmse(), L.R2Score()
    a.reset(); b.reset()
    ps, ts = [], []
    for _ in range(4):
        p, t = torch.randn(2, 5, 5, generator=g), torch.randn(2, 5, 5, generator=g) + 0.3
        a.accumulate_values(p, t); b.accumulate_values(p, t)
        ps.append(p); ts.append(t)
    P, T = torch.cat(ps), torch.cat(ts)
    assert abs(a.value - O.rmse(P, T)) < 1e-9 and abs(b.value - O.r2_score(P, T)) < 1e-9
    assert (a.name, b.name) == ("_rmse", "r2_score")          # fastai metric names (monitor 'r2_score', train.py:199)
    from sklearn.metrics import r2_score
    assert abs(b.value - r2_score(T.reshape(-1).numpy(), P.reshape(-1).numpy())) < 1e-6


def test_lr_suggestions_on_a_synthetic_sweep():
    lrs = np.logspace(-7, 1, 100)
    x = np.log10(lrs)
    losses = 2.0 - 1.5 / (1 + np.exp(-(x + 4) * 2)) + np.where(x > -1.5, (x + 1.5) ** 2 * 2, 0)     # plateau, descent, blow-up
    v, st, mn, sl = (L.lr_valley(lrs, losses, 100), L.lr_steep(lrs, losses, 100), L.lr_minimum(lrs, losses, 100),
                     L.lr_slide(lrs, losses, 100))
    lo = float(lrs[np.argmin(losses)])
    assert 1e-6 < v < lo                       # valley: inside the descending stretch, left of the minimum
    assert abs(np.log10(st) + 4) < 0.3         # steepest descent of the sigmoid at 1e-4
    assert abs(mn - lo / 10) < 1e-12
    assert 1e-7 <= sl <= 10


def test_tile_dataset_float_targets_for_regression():
    x = np.zeros((2, 4, 4), dtype=np.uint8)
    m = np.arange(16, dtype=np.float32).reshape(4, 4) / 3
    ds = L.TileDataset([x], [m], "int8", regression=True)
    _, y = ds[0]
    assert y.dtype == torch.float32 and torch.allclose(y, torch.from_numpy(m))
    _, y2 = L.TileDataset([x], [m], "int8")[0]
    assert y2.dtype == torch.int64


def test_dataloader_rank_sharding():
    """tile-DDP: same permutation on every rank, disjoint strided shards, equal step counts for the (shuffled) training loader,
    every item exactly once for a validation loader"""
    imgs = [np.full((1, 4, 4), i, dtype=np.uint8) for i in range(11)]
    masks = [np.full((4, 4), i, dtype=np.uint8) for i in range(11)]
    ds = L.TileDataset(imgs, masks, "int8")
    seen = []
    for r in range(2):
        dl = L.DataLoader(ds, 2, True, "cpu", drop_last=True, seed=5).shard(r, 2)
        assert len(dl) == 2                       # 11 // 2 = 5 items per rank -> 2 full batches
        seen.append([int(v) for _, yb in dl for v in yb[:, 0, 0]])
    assert len(seen[0]) == len(seen[1]) == 4 and not set(seen[0]) & set(seen[1])
    val = [sorted(int(v) for _, yb in L.DataLoader(ds, 4, False, "cpu").shard(r, 3) for v in yb[:, 0, 0]) for r in range(3)]
    assert sorted(sum(val, [])) == list(range(11)) and val[0] == [0, 3, 6, 9]


def test_augmentation_pipeline_has_albumentations_semantics():
    """unet_amd.augment: Compose order / probabilities, RandomBrightnessContrast (img * alpha + beta, clipped to [0,1], mask untouched),
    CoarseDropout (8 holes of 8 x 8 by default, image only), and the reference's batch slicing rule"""
    from unet_amd import augment as A
    g = np.random.default_rng(0)
    img = torch.rand(4, 32, 40)
    mask = torch.randint(0, 3, (32, 40))
    o, m = A.RandomBrightnessContrast(brightness_limit=(0.1, 0.1), contrast_limit=(0.2, 0.2), p=1.0)(img, mask, g)
    assert torch.allclose(o, (img * 1.2 + 0.1).clamp(0, 1)) and m is mask
    o, _ = A.RandomBrightnessContrast(brightness_limit=(0.1, 0.1), contrast_limit=(0.0, 0.0), brightness_by_max=False, p=1.0)(img, mask, g)
    assert torch.allclose(o, (img + 0.1 * img.mean()).clamp(0, 1))
    o, m = A.CoarseDropout(p=1.0)(torch.ones(4, 64, 64), mask, g)
    holes = (o[0] == 0)
    assert bool((o == 0).all(0).eq(holes).all()) and 8 * 8 <= int(holes.sum()) <= 8 * 64 and m is mask     # all bands, image only
    o, m2 = A.CoarseDropout(max_holes=3, max_height=4, max_width=6, min_holes=1, min_height=2, min_width=2, fill_value=0.5,
                            mask_fill_value=9, p=1.0)(torch.ones(2, 20, 20), torch.zeros(20, 20, dtype=torch.long), g)
    assert bool(((o[0] == 0.5) == (m2 == 9)).all()) and 4 <= int((m2 == 9).sum()) <= 72
    # Compose applies in order; p = 0 members never fire
    pipe = A.Compose([A.HorizontalFlip(p=1.0), A.VerticalFlip(p=0.0), A.RandomBrightnessContrast(p=0.0)])
    o, m = pipe(img, mask, g)
    assert torch.equal(o, img.flip(-1)) and torch.equal(m, mask.flip(-1))
    # batch rule: ceil(B * n) - B candidates, counted from the front
    xb, yb = torch.rand(6, 4, 8, 8), torch.randint(0, 3, (6, 8, 8))
    x0 = xb.clone()
    A.BatchAugment(A.Compose([A.HorizontalFlip(p=1.0)]), n_transform_imgs=0.5)(xb, yb)
    assert torch.equal(xb[:3], x0[:3].flip(-1)) and torch.equal(xb[3:], x0[3:])
    A.BatchAugment(A.Compose([A.HorizontalFlip(p=1.0)]), n_transform_imgs=1.0)(xb, yb)       # quirk Q7: nothing
    assert torch.equal(xb[:3], x0[:3].flip(-1))
    with pytest.raises(ValueError):
        A.BatchAugment(A.default_pipeline(), n_transform_imgs=1.5)
