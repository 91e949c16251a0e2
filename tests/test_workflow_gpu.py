"""End-to-end tile workflow on the GPU: train_func (25 positional arguments) -> export -> save_predictions with and
without overlap merge, on synthetic GeoTIFF tiles written by unet_amd.tiffio."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_dataset(root, n_train=6, n_val=2, size=64, n_cls=3, seed=0):
    from unet_amd.tiffio import write_tiff
    g = np.random.default_rng(seed)
    for split, n in (("trai", n_train), ("vali", n_val)):
        (root / split / "img_tiles").mkdir(parents=True)
        (root / split / "mask_tiles").mkdir(parents=True)
        for i in range(n):
            mask = np.zeros((size, size), dtype=np.uint8)
            mask[:, size // 3:] = 1
            mask[size // 2:, 2 * size // 3:] = 2
            img = (mask[None].astype(np.float32) * 80 + g.normal(40, 10, (4, size, size))).clip(0, 255).astype(np.uint8)
            gt = (500000.0 + i * size * 0.2, 0.2, 0.0, 5800000.0, 0.0, -0.2)
            write_tiff(root / split / "img_tiles" / f"t{i}.tif", img, geotransform=gt)
            write_tiff(root / split / "mask_tiles" / f"t{i}.tif", mask, geotransform=gt)


def test_train_export_predict_merge(tmp_path):
    import train as T
    import predict as P
    from unet_amd import xresnet18
    from unet_amd.tiffio import read_tiff, write_tiff
    data = tmp_path / "data"
    _make_dataset(data)
    codes = ["a", "b", "c"]
    learn = T.train_func(data, None, tmp_path / "models", "run1", 2, False, False, "weighted", xresnet18, 3, 2e-3, 10, None, None,
                         "dice_multi", False, ["vali"], codes, False, None, True, None, 1, "", False)
    hist = (tmp_path / "models" / "run1" / "run1_history.csv").read_text().strip().splitlines()
    assert hist[0] == "epoch,train_loss,valid_loss,dice_multi,time" and len(hist) == 4
    first, last = float(hist[1].split(",")[2]), float(hist[-1].split(",")[2])
    assert np.isfinite(last) and last < first * 1.5
    pkl = tmp_path / "models" / "run1" / "run1.pkl"
    assert pkl.exists() and (tmp_path / "models" / "run1" / "run1_model_summary.txt").exists()

    # prediction tiles with 50 % horizontal overlap
    pred = tmp_path / "pred" / "tiles"
    pred.mkdir(parents=True)
    g = np.random.default_rng(1)
    for i in range(3):
        img = g.integers(0, 255, (4, 64, 64)).astype(np.uint8)
        write_tiff(pred / f"p{i}.tif", img, geotransform=(1000.0 + i * 32 * 0.5, 0.5, 0.0, 2000.0, 0.0, -0.5))
    out_dir = P.save_predictions(pkl, pred, False, merge=False, validation_vision=False)
    m0, meta = read_tiff(out_dir / "p0.tif")
    assert m0.shape == (64, 64) and m0.dtype == np.uint8 and meta["geotransform"][0] == 1000.0
    merged_file = P.save_predictions(pkl, pred, False, merge=True, AOI="aoi", year="2024", validation_vision=False)
    mm, meta = read_tiff(merged_file)
    assert mm.shape == (64, 128) and meta["geotransform"][:2] == (1000.0, 0.5)
    # the merged mask equals sum-probs / count / argmax computed on the host from per-tile probabilities
    lr = T.load_learner(pkl)
    acc = np.zeros((3, 64, 128)); cnt = np.zeros((64, 128))
    for i in range(3):
        _, _, pr = lr.predict(pred / f"p{i}.tif")
        acc[:, :, i * 32:i * 32 + 64] += pr.numpy(); cnt[:, i * 32:i * 32 + 64] += 1
    assert np.array_equal((acc / cnt).argmax(0).astype(np.uint8), mm)
    # non-overlapped columns of the merge equal the per-tile masks
    assert np.array_equal(mm[:, :32], m0[:, :32])
    # tiles of MIXED sizes in one folder (per-tile outputs): the feeder-backed prefetcher needs one shape, such a set keeps the general one;
    # every tile's mask equals the mask of the tile predicted alone
    mixed = tmp_path / "pred" / "mixed"
    mixed.mkdir()
    shapes = [(64, 64), (64, 64), (96, 64), (64, 64), (96, 64)]
    for i, (h, w) in enumerate(shapes):
        write_tiff(mixed / f"q{i}.tif", g.integers(0, 255, (4, h, w)).astype(np.uint8), geotransform=(3000.0 + 100 * i, 0.5, 0.0, 2000.0, 0.0, -0.5))
    assert isinstance(P._prefetcher([mixed / f"q{i}.tif" for i in range(5)], [(0, 2), (2, 1)], shapes[:3], torch.device("cuda")), P._TilePrefetcher)
    assert isinstance(P._prefetcher([mixed / f"q{i}.tif" for i in range(2)], [(0, 2)], shapes[:2], torch.device("cuda")), P._FeedPrefetcher)
    out_mixed = P.save_predictions(pkl, mixed, False, merge=False, validation_vision=False, batch_size=2)
    for i, (h, w) in enumerate(shapes):
        mi, _ = read_tiff(out_mixed / f"q{i}.tif")
        assert mi.shape == (h, w)
        dec, _, _ = lr.predict(mixed / f"q{i}.tif")
        assert np.array_equal(mi, dec.numpy().astype(np.uint8)), i


def _make_regression_dataset(root, n_train=6, n_val=2, size=64, seed=0):
    from unet_amd.tiffio import write_tiff
    g = np.random.default_rng(seed)
    for split, n in (("trai", n_train), ("vali", n_val)):
        (root / split / "img_tiles").mkdir(parents=True)
        (root / split / "mask_tiles").mkdir(parents=True)
        for i in range(n):
            img = g.integers(0, 255, (4, size, size)).astype(np.uint8)
            target = (img[0].astype(np.float32) + img[1]) / 100.0          # a smooth function of the inputs
            gt = (500000.0 + i * size * 0.2, 0.2, 0.0, 5800000.0, 0.0, -0.2)
            write_tiff(root / split / "img_tiles" / f"t{i}.tif", img, geotransform=gt)
            write_tiff(root / split / "mask_tiles" / f"t{i}.tif", target, geotransform=gt)


def test_regression_workflow_with_lr_finder(tmp_path):
    """enable_regression = True and LR_FINDER = 'valley' through train_func -> export -> load_learner -> predict (2-tuple)
    -> save_predictions with overlap merge (mean of overlapping tiles, nodata -9999)"""
    import train as T
    import predict as P
    from unet_amd import xresnet18
    from unet_amd.learner import Learner_adjust
    from unet_amd.tiffio import read_tiff, write_tiff
    data = tmp_path / "data"
    _make_regression_dataset(data)
    learn = T.train_func(data, None, tmp_path / "models", "reg", 2, False, True, "even", xresnet18, 2, 1e-3, 10, "valley", None,
                         None, False, ["vali"], ["value"], False, None, True, None, 1, "", False)
    assert isinstance(learn, Learner_adjust) and learn.model.n_out == 1
    hist = (tmp_path / "models" / "reg" / "reg_history.csv").read_text().strip().splitlines()
    assert hist[0] == "epoch,train_loss,valid_loss,_rmse,r2_score,time" and len(hist) == 3
    row = [float(v) for v in hist[-1].split(",")[1:5]]
    assert all(np.isfinite(row)) and abs(row[2] - np.sqrt(row[1])) < 1e-3 * max(1.0, row[2])      # rmse = sqrt(valid MSE)
    # the LR finder restored the weights it started from (fit_one_cycle then trained from the initial state) and left a curve
    assert len(learn.lr_find_curve[0]) >= 6
    pkl = tmp_path / "models" / "reg" / "reg.pkl"
    lr = T.load_learner(pkl)
    assert isinstance(lr, Learner_adjust)
    pred = tmp_path / "pred" / "tiles"
    pred.mkdir(parents=True)
    g = np.random.default_rng(1)
    for i in range(2):
        write_tiff(pred / f"p{i}.tif", g.integers(0, 255, (4, 64, 64)).astype(np.uint8),
                   geotransform=(1000.0 + i * 96 * 0.5, 0.5, 0.0, 2000.0, 0.0, -0.5))       # a 32-pixel gap between the tiles
    dec, pr = lr.predict(pred / "p0.tif")
    assert dec.shape == (1, 64, 64) and torch.equal(dec, pr)
    out_dir = P.save_predictions(pkl, pred, True, merge=False, validation_vision=False)
    t0, _ = read_tiff(out_dir / "p0.tif")
    assert t0.dtype == np.float32 and np.allclose(t0.reshape(64, 64), pr[0].numpy(), atol=1e-6)
    merged = P.save_predictions(pkl, pred, True, merge=True, AOI="aoi", validation_vision=False)
    mm, meta = read_tiff(merged / "aoi_reg_prediction.tif")
    assert mm.shape == (64, 160) and meta.get("nodata") == -9999.0
    assert np.all(mm[:, 64:96] == -9999) and np.allclose(mm[:, :64], t0.reshape(64, 64), atol=1e-6)


def _train_worker(rank, world, port, data, models, q):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UNET_DIST_BACKEND="gloo", UNET_FORCE_DEVICE="0")
    import torch.distributed as dist
    import train as T
    from unet_amd import xresnet18
    torch.manual_seed(500 + rank)                 # different initial weights per rank: train_func must end with identical replicas
    learn = T.train_func(data, None, models, "ddp", 2, False, False, "weighted", xresnet18, 2, 2e-3, 10, None, None,
                         "dice_multi", False, ["vali"], ["a", "b", "c"], True, None, True, None, 0.5, "", False)
    p = learn.model.flat_param.clone()
    ref = p.clone()
    dist.broadcast(ref, 0)
    q.put((rank, bool(torch.equal(p, ref)), learn.world, len(learn.dls.train)))
    dist.barrier()
    dist.destroy_process_group()


def test_train_func_under_two_ranks(tmp_path):
    """train_func launched once per GPU (here: two ranks sharing the one GPU over gloo): the training tiles are sharded (8 tiles,
    batch 2 -> 2 steps per epoch and rank instead of 4), replicas end identical, and exactly ONE set of result files exists"""
    import socket
    import torch.multiprocessing as mp
    data = tmp_path / "data"
    _make_dataset(data, n_train=8, n_val=2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, str(data), str(tmp_path / "models"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(0, True, 2, 2), (1, True, 2, 2)], res
    out = tmp_path / "models" / "ddp"
    hist = (out / "ddp_history.csv").read_text().strip().splitlines()
    assert hist[0] == "epoch,train_loss,valid_loss,dice_multi,time" and len(hist) == 3
    assert (out / "ddp.pkl").exists() and (out / "ddp_model_summary.txt").exists() and (out / "ddp.json").exists()
    assert '"world_size": 2' in (out / "ddp.json").read_text()
    assert not (out / "history.csv").exists()
    import train as T
    lr = T.load_learner(out / "ddp.pkl")
    assert lr.model.n_out == 3
