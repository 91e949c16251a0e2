"""The training feed (unet_amd/feed.py + unet_tiles_stage / unet_mask_stage / unet_dice_counts) against the host path it replaces.

The host path IS the reference's order of work (train.py:345 -> fastai DataLoader(num_workers=0) -> data.py:18-28 open_npy per item ->
utils.py:239-295 batch transform: / 255 for int8 data, / 255 twice for int16 data, flips on the first ceil(B * n_transform_imgs) - B
images): the device feed must hand the step the same (xb, yb) bit for bit and a fit must produce the same losses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiles(n, n_in, size, dtype, seed, n_cls=4):
    g = np.random.default_rng(seed)
    hi = 256 if dtype == np.uint8 else 60000
    imgs = [g.integers(0, hi, (n_in, *size)).astype(dtype) for _ in range(n)]
    masks = [g.integers(0, n_cls, size).astype(np.uint8) for _ in range(n)]
    return imgs, masks


def _pair(ds, bs, tfm_a, tfm_b, shuffle=True, seed=5, **kw):
    from unet_amd.learner import DataLoader
    a = DataLoader(ds, bs, shuffle, "cuda", seed=seed, batch_tfm=tfm_a, feed="host")
    b = DataLoader(ds, bs, shuffle, "cuda", seed=seed, batch_tfm=tfm_b, feed="device", **kw)
    return a, b


@pytest.mark.parametrize("sample,dtype", [(np.uint8, "int8"), (np.uint16, "int16"), (np.int16, "int16"), (np.int32, "int8"), (np.float32, "int8")])
def test_device_feed_equals_host_path_bit_for_bit(sample, dtype):
    """in-memory tiles of every sample type the kernels read, the int16 rule (/ 255 twice), a ragged last batch, two epochs (the shuffle
    and the flip draws advance identically), flips on the first ceil(B * 0.5) - B images of every batch"""
    from unet_amd.learner import FlipAugment, TileDataset
    n, bs = 11, 4
    imgs, masks = _tiles(n, 4, (40, 56), np.uint8 if sample == np.float32 else sample, 1)
    if sample == np.float32:
        imgs = [a.astype(np.float32) + 0.75 for a in imgs]         # data.py:24 casts through int32: the fraction is dropped on both paths
    if sample == np.int16:
        imgs = [(a.astype(np.int32) - 20000).astype(np.int16) for a in imgs]
    ds = TileDataset(imgs, masks, dtype)
    host, dev = _pair(ds, bs, FlipAugment(n_transform_imgs=0.5, seed=3), FlipAugment(n_transform_imgs=0.5, seed=3))
    flipped = 0
    for _ in range(2):
        got, want = list(dev), list(host)
        assert len(got) == len(want) == 3
        for (xa, ya), (xb, yb) in zip(want, got):
            assert xb.is_cuda and xb.dtype == torch.float32 and yb.dtype == torch.int64 and xb.shape == xa.shape
            assert torch.equal(xa, xb) and torch.equal(ya, yb)
        flipped += sum(int(not torch.equal(xb[0], torch.from_numpy(np.stack(imgs)).cuda()[0])) for xb, _ in got)
    assert flipped > 0


def test_device_feed_reads_tile_files_like_the_host_path(tmp_path):
    """GeoTIFF tile files (uncompressed from write_tiff, LZW from libtiff when Pillow is there), .npy tiles, a validation loader without
    transform, a prediction loader without masks; more workers than items; depth 1"""
    from unet_amd.learner import DataLoader, TileDataset
    from unet_amd.tiffio import write_tiff
    imgs, masks = _tiles(7, 4, (64, 48), np.uint8, 2)
    pi, pm = [], []
    for i, (a, m) in enumerate(zip(imgs, masks)):
        if i % 3 == 2:
            np.save(tmp_path / f"i{i}.npy", a); np.save(tmp_path / f"m{i}.npy", m)
            pi.append(tmp_path / f"i{i}.npy"); pm.append(tmp_path / f"m{i}.npy")
            continue
        done = False
        if i % 3 == 1:
            try:
                from PIL import Image
                Image.fromarray(np.moveaxis(a, 0, -1), "RGBA").save(tmp_path / f"i{i}.tif", compression="tiff_lzw")
                Image.fromarray(m, "L").save(tmp_path / f"m{i}.tif", compression="tiff_lzw")
                done = True
            except ImportError:
                pass
        if not done:
            write_tiff(tmp_path / f"i{i}.tif", a); write_tiff(tmp_path / f"m{i}.tif", m)
        pi.append(tmp_path / f"i{i}.tif"); pm.append(tmp_path / f"m{i}.tif")
    ds = TileDataset(pi, pm, "int8")
    host, dev = _pair(ds, 3, None, None, shuffle=False, workers=12, depth=1)
    for (xa, ya), (xb, yb) in zip(list(host), list(dev)):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    x0 = torch.cat([xb for xb, _ in dev]).cpu()
    assert torch.equal(x0, torch.from_numpy(np.stack(imgs).astype(np.int32).astype(np.float32) / 255.0))
    test_dl = DataLoader(TileDataset(pi, None, "int8"), 4, False, "cuda")
    out = list(test_dl)
    assert all(yb is None for _, yb in out) and torch.equal(torch.cat([xb for xb, _ in out]).cpu(), x0)


def test_jpeg_compressed_tile_files_through_either_feed(tmp_path):
    """COMPRESS=JPEG tiles (4 bands, what GDAL writes for orthophoto tiles; masks stay lossless): the device feed and the host path stage the
    same integers -- the ones libtiff decodes (tests/test_tiff_jpeg_cpu.py has the decoder's own bars)"""
    Image = pytest.importorskip("PIL.Image")
    from unet_amd.learner import TileDataset
    from unet_amd.tiffio import write_tiff
    imgs, masks = _tiles(6, 4, (64, 48), np.uint8, 2)
    pi, pm, ref = [], [], []
    for i, (a, m) in enumerate(zip(imgs, masks)):
        Image.fromarray(np.moveaxis(a, 0, -1), "RGBA").save(tmp_path / f"i{i}.tif", compression="jpeg", quality=90)
        ref.append(np.moveaxis(np.asarray(Image.open(tmp_path / f"i{i}.tif")), -1, 0))
        write_tiff(tmp_path / f"m{i}.tif", m)
        pi.append(tmp_path / f"i{i}.tif"); pm.append(tmp_path / f"m{i}.tif")
    host, dev = _pair(TileDataset(pi, pm, "int8"), 3, None, None, shuffle=False, workers=4, depth=2)
    for (xa, ya), (xb, yb) in zip(list(host), list(dev)):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    x0 = torch.cat([xb for xb, _ in dev]).cpu()
    assert torch.equal(x0, torch.from_numpy(np.stack(ref).astype(np.int32).astype(np.float32) / 255.0))


def test_single_batch_loaders_stage_directly_and_equal_the_host_path():
    """feed="auto" with ONE batch (Learner.predict on a tile, a tiny validation set): no thread pool, no staging ring -- the integers go up in
    one copy and the same kernels scale / widen / flip them: bit-equal to the host path, for uint16 data (/ 255 twice), flips, float targets"""
    from unet_amd.learner import DataLoader, FlipAugment, TileDataset
    imgs, masks = _tiles(3, 4, (40, 56), np.uint16, 8)
    for regression in (False, True):
        mk = [m.astype(np.float32) * 0.37 for m in masks] if regression else masks
        ds = TileDataset(imgs, mk, "int16", regression=regression)
        host = DataLoader(ds, 4, False, "cuda", batch_tfm=FlipAugment(n_transform_imgs=0.3, seed=5), feed="host")
        auto = DataLoader(ds, 4, False, "cuda", batch_tfm=FlipAugment(n_transform_imgs=0.3, seed=5))
        (xa, ya), = list(host)
        (xb, yb), = list(auto)
        assert auto._feeder is None                         # the direct path: no pool was built
        assert torch.equal(xa, xb) and torch.equal(ya, yb) and yb.dtype == (torch.float32 if regression else torch.int64)
    x1, y1 = next(iter(DataLoader(TileDataset(imgs[:1], None, "int16"), 16, False, "cuda")))
    assert y1 is None and torch.equal(x1.cpu(), torch.from_numpy(imgs[0].astype(np.int32).astype(np.float32) / 255.0 / 255.0)[None])


def test_regression_targets_and_generic_pipelines_go_through_the_device_feed():
    """float mask tiles (RegressionBlock, data.py:98-99) arrive as float32 targets; a pipeline that is not made of flips only
    (RandomBrightnessContrast) runs as torch ops on the staged batch, with the same draws as on the host path; a flips-only
    albumentations-style pipeline (the reference's default aug_pipe) is folded into the staging kernels"""
    from unet_amd import augment as A
    from unet_amd.learner import TileDataset
    imgs, _ = _tiles(6, 3, (32, 32), np.uint8, 4)
    g = np.random.default_rng(9)
    targs = [g.normal(size=(32, 32)).astype(np.float64 if i % 2 else np.float32) for i in range(6)]
    ds = TileDataset(imgs, [t.astype(np.float32) for t in targs], "int8", regression=True)
    mk = lambda: A.BatchAugment(A.default_pipeline(), n_transform_imgs=0.4, seed=11)
    assert hasattr(mk(), "flip_flags")
    host, dev = _pair(ds, 3, mk(), mk())
    for (xa, ya), (xb, yb) in zip(list(host), list(dev)):
        assert yb.dtype == torch.float32 and torch.equal(xa, xb) and torch.equal(ya, yb)
    mk2 = lambda: A.BatchAugment(A.Compose([A.HorizontalFlip(p=0.5), A.RandomBrightnessContrast(p=1.0)]), n_transform_imgs=0.4, seed=12)
    assert not hasattr(mk2(), "flip_flags")
    host, dev = _pair(ds, 3, mk2(), mk2())
    for (xa, ya), (xb, yb) in zip(list(host), list(dev)):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)


def test_three_steps_of_fit_give_the_same_losses_through_either_feed(tmp_path):
    """Learner.fit_one_cycle over the same tiles through the host path and through the device feed: identical smoothed losses, identical
    parameters afterwards (same kernels on bit-equal inputs); validation loss and DiceMulti equal (counters from unet_dice_counts)"""
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, FlipAugment, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    imgs, masks = _tiles(6, 4, (64, 64), np.uint8, 6, n_cls=3)
    res = []
    for feed in ("host", "device"):
        torch.manual_seed(3)
        model = HipDynamicUnet("xresnet18", 4, 3, (64, 64))
        dls = DataLoaders(TileDataset(imgs, masks, "int8"), TileDataset(imgs[:3], masks[:3], "int8"), 2, vocab=list("abc"), seed=7,
                          train_tfm=FlipAugment(n_transform_imgs=0.5, seed=2), feed=feed)
        learn = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1, weight=torch.tensor([0.2, 0.3, 0.5])), metrics=[DiceMulti()],
                        path=tmp_path)
        learn._no_logging = True
        learn.fit_one_cycle(1, lr_max=slice(2e-4, 2e-3))
        torch.cuda.synchronize()
        res.append((list(learn.recorder.losses), list(learn.recorder.values[-1]), model.flat_param.detach().clone()))
    (la, va, pa), (lb, vb, pb) = res
    assert len(la) == 3 and la == lb, (la, lb)
    assert va == vb, (va, vb)
    assert torch.equal(pa, pb)


def test_dice_counts_kernel_against_bincount():
    from unet_amd import ops
    g = torch.Generator().manual_seed(0)
    for n_cls, P in ((5, 70001), (2, 513), (64, 300000)):
        p = torch.randint(0, n_cls, (P,), generator=g)
        t = torch.randint(-1, n_cls + 2, (P,), generator=g)          # out-of-range targets are clamped for the target count only
        counts = torch.zeros((3, n_cls), dtype=torch.int64, device="cuda")
        ops.dice_counts(p.cuda(), t.cuda(), n_cls, counts)
        ops.dice_counts(p.cuda(), t.cuda(), n_cls, counts)          # accumulates
        c = counts.cpu()
        assert torch.equal(c[0], 2 * torch.bincount(p[p == t], minlength=n_cls)[:n_cls])
        assert torch.equal(c[1], 2 * torch.bincount(p, minlength=n_cls))
        assert torch.equal(c[2], 2 * torch.bincount(t.clamp(0, n_cls - 1), minlength=n_cls))


def test_stage_kernels_reject_what_they_cannot_do():
    from unet_amd import _lib as L
    src = torch.zeros((1, 4, 8, 8), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((1, 4, 8, 8), dtype=torch.float32, device="cuda")
    assert L.lib.unet_tiles_stage(src.data_ptr(), 0, 65, 4, 8, 8, 0, 0, 0, dst.data_ptr(), None) == -1        # > 64 images per call
    assert L.lib.unet_tiles_stage(src.data_ptr(), 9, 1, 4, 8, 8, 0, 0, 0, dst.data_ptr(), None) == -1         # unknown sample type
    assert L.lib.unet_mask_stage(None, 0, 1, 8, 8, 0, 0, dst.data_ptr(), 0, None) == -1
    assert L.lib.unet_dice_counts(src.data_ptr(), src.data_ptr(), 10, 65, dst.data_ptr(), None) == -1
