"""BASELINE.json configs[4] as a product path: predict.predict_raster (sliding-window inference over a raster resident in HBM).

  (i)   a ~700 px raster with ragged last windows, a nodata value and an empty corner: predict_raster == split_raster -> tile files ->
        save_predictions(merge=True) bit for bit (reference flow create_tiles_unet.py:252-434 -> predict.py:146-334), and == the CPU
        oracle's cut -> predict -> sum / count -> argmax within the stated float tolerance (masks: identical outside fp32 ties);
  (ii)  the device kernels against the per-tile path (scale_input -> predict_probs -> unet_mosaic_accumulate) bit for bit, for uint8 / uint16
        rasters, bf16 storage and regression;
  (iii) the full 20000 x 20000 run (2401 windows of 512, overlap 0.2) through size-independent properties;
  (iv)  two ranks (gloo, one GPU): the row-block partition with slab exchange equals one rank bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)


def _pair(arch, n_in, n_out, size, seed, act_dtype="f32", head_target=4.0):
    """(HIP model, oracle) with identical weights, eval mode, logits normalised to O(1)"""
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(seed)
    ref = O.DynamicUnet(arch, n_in, n_out, (size, size))
    ref.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randint(0, 256, (1, n_in, size, size), generator=g).float() / 255
    with torch.no_grad():
        for m in ref.modules():                      # non-trivial running statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.05, generator=g)
                m.running_var.uniform_(0.7, 1.3, generator=g)
                m.weight.add_(torch.randn(m.weight.shape, generator=g) * 0.1)
        s = ref(x).abs().max().item() / head_target
        head = ref.layers[-1][0]
        head.weight.div_(s)
        head.bias.div_(s)
    model = HipDynamicUnet(arch, n_in, n_out, (size, size), act_dtype=act_dtype)
    r = model.load_state_dict(ref.state_dict())
    assert not r.missing_keys and not r.unexpected_keys
    model.eval()
    return model, ref


def _raster(seed, C, H, W, dtype=np.uint8, hi=256):
    g = np.random.default_rng(seed)
    return g.integers(1, hi, (C, H, W)).astype(dtype)


def _reference_windows(img, size, overlap, max_empty, nodata):
    """create_tiles_unet.py:344-379 restated in numpy: nodata -> 0, windows, emptiness filter.  Returns (image, [(y, x)])"""
    from unet_amd.mosaic import sliding_windows
    img = img.copy()
    if nodata is not None:
        img[:, (img == nodata).any(axis=0)] = 0
    keep = []
    for y, x in sliding_windows(img.shape[1], img.shape[2], size, overlap):
        crop = np.moveaxis(img[:, y:y + size, x:x + size], 0, 2)
        if np.sum(crop != 0) < np.prod(crop.shape) * (1 - max_empty):
            continue
        keep.append((int(y), int(x)))
    return img, keep


def _tile_path_merge(model, img, wins, size, dtype, C_out, regression=False, rep=1):
    """the per-tile device path the engine must reproduce bit for bit: scale_input -> predict_probs -> unet_mosaic_accumulate per tile.
    rep: the tile is presented `rep` times, i.e. in the batch geometry the engine runs (the fp32 conv planner picks its tile -- and with
    it the 4x4x1 sliver for a 16 n + 1..4 wide layer, whose two accumulation chains round differently -- by the launch's size)"""
    from unet_amd import ops
    from unet_amd.learner import scale_input
    oy, ox = min(w[0] for w in wins), min(w[1] for w in wins)
    MH, MW = max(w[0] for w in wins) + size - oy, max(w[1] for w in wins) + size - ox
    mosaic = torch.zeros((C_out, MH, MW), dtype=torch.float32, device="cuda")
    count = torch.zeros((MH, MW), dtype=torch.int32, device="cuda")
    for y, x in wins:
        t = torch.from_numpy(scale_input(img[:, y:y + size, x:x + size], dtype))[None].cuda().repeat(rep, 1, 1, 1)
        p = model.predict_values(t) if regression else model.predict_probs(t)[0]
        ops.mosaic_accumulate(p[0].contiguous(), mosaic, count, y - oy, x - ox)
    return mosaic, count


def test_predict_raster_equals_tile_files_flow_and_oracle(tmp_path):
    import create_tiles_unet as T
    import predict as P
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, Learner, TileDataset, scale_input
    from unet_amd.tiffio import read_tiff, write_tiff
    size, overlap, max_empty, nodata = 256, 0.2, 0.9, 250
    model, ref = _pair("xresnet18", 4, 3, size, seed=11)
    img = _raster(5, 4, 700, 620, hi=250)
    img[:, :250, :250] = 0                    # window (0, 0) is > 90 % empty: dropped, part of its area stays uncovered
    img[1, 300:303, 100:400] = nodata         # nodata in one band zeroes the pixel in every band
    gt = (400000.0, 0.5, 0.0, 5700000.0, 0.0, -0.5)
    rpath = tmp_path / "scene.tif"
    write_tiff(rpath, img, geotransform=gt, nodata=nodata)
    dls = DataLoaders(TileDataset([np.zeros((4, size, size), np.uint8)], None, "int8"), None, 1, device="cuda", vocab=["a", "b", "c"])
    learn = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), metrics=[DiceMulti()], path=tmp_path)
    pkl = tmp_path / "m.pkl"
    learn.export(pkl)

    tm = {}
    out = P.predict_raster(model, rpath, size, overlap, max_empty=max_empty, batch_size=5, out_path=tmp_path / "direct.tif", timing=tm)
    allc = P.predict_raster(model, rpath, size, overlap, max_empty=max_empty, batch_size=5, all_classes=True)
    zimg, wins = _reference_windows(img, size, overlap, max_empty, nodata)
    assert tm["all_windows"] == 12 and tm["kept_windows"] == len(wins) == 11 and (0, 0) not in wins
    hits = np.zeros((700, 620), np.int32)
    for y, x in wins:
        hits[y:y + size, x:x + size] += 1
    assert out.dtype == np.uint8 and out.shape == (700, 620) == allc.shape[1:]
    assert tm["hits_min"] == hits.min() == 0 and tm["hits_max"] == hits.max() == 6      # three window rows x two columns meet near the ragged end

    # (a) the reference's two-step flow through tile files
    tiles = tmp_path / "cut"
    T.split_raster(rpath, None, tiles, patch_size=size, patch_overlap=overlap, split=[1], max_empty=max_empty)
    assert len(list((tiles / "img_tiles").glob("*.tif"))) == len(wins)
    f = P.save_predictions(pkl, tiles / "img_tiles", False, merge=True, AOI="flow", validation_vision=False, batch_size=5)
    via_files, meta = read_tiff(f)
    assert np.array_equal(via_files, out)
    d, dmeta = read_tiff(tmp_path / "direct.tif")
    assert np.array_equal(d, out) and dmeta["geotransform"] == meta["geotransform"] == gt
    f2 = P.save_predictions(pkl, tiles / "img_tiles", False, merge=True, all_classes=True, AOI="flowall", validation_vision=False, batch_size=5)
    assert np.array_equal(read_tiff(f2)[0], allc)                 # float mosaics: same additions in the same order
    # the reference's int8 "large_file" merge (predict.py:209-214, 324-329), raster path == tile-file path == the host arithmetic in numpy
    f3 = P.save_predictions(pkl, tiles / "img_tiles", False, merge=True, all_classes=True, large_file=True, AOI="flow8", validation_vision=False,
                            batch_size=5)
    big8 = P.predict_raster(model, rpath, size, overlap, max_empty=max_empty, batch_size=5, all_classes=True, large_file=True)
    assert big8.dtype == np.int8 and np.array_equal(read_tiff(f3)[0], big8)

    # (b) the per-tile device path
    mosaic, count = _tile_path_merge(model, zimg, wins, size, "int8", 3, rep=5)
    from unet_amd import ops
    am = torch.empty(count.shape, dtype=torch.uint8, device="cuda")
    ops.mosaic_finalize(mosaic, count, am)
    assert np.array_equal(am.cpu().numpy(), out) and np.array_equal(mosaic.cpu().numpy(), allc)

    acc8, cnt8 = np.zeros((3, 700, 620), np.int8), np.zeros((3, 700, 620), np.int8)
    for y, x in wins:
        t = torch.from_numpy(scale_input(zimg[:, y:y + size, x:x + size], "int8"))[None].cuda().repeat(5, 1, 1, 1)
        pr = model.predict_probs(t)[0][0].cpu().numpy()
        acc8[:, y:y + size, x:x + size] += np.around(pr * 31).astype(np.int8)
        cnt8[:, y:y + size, x:x + size] += 1
    acc8[cnt8 > 0] //= cnt8[cnt8 > 0]
    assert np.array_equal(acc8, big8)

    # (c) the CPU oracle: cut -> predict -> sum of probabilities / hit counter -> argmax (predict.py:193-203, 284-334)
    acc, cnt = np.zeros((3, 700, 620), np.float32), np.zeros((700, 620), np.int32)
    with torch.no_grad():
        for y, x in wins:
            t = torch.from_numpy(scale_input(zimg[:, y:y + size, x:x + size], "int8"))[None]
            pr = torch.softmax(ref(t), dim=1)[0].numpy()
            acc[:, y:y + size, x:x + size] += pr
            cnt[y:y + size, x:x + size] += 1
    acc[:, cnt > 0] /= cnt[cnt > 0]
    assert np.array_equal(cnt > 0, count.cpu().numpy() > 0)
    err = np.abs(acc - allc).max()
    assert err < 1e-4, err                                         # probabilities of O(1) logits: fp32 rounding of two implementations
    diff = acc.argmax(0) != out
    top2 = np.sort(acc, axis=0)[-2:]
    if diff.any():
        print(f"{int(diff.sum())} mask pixel(s) differ; oracle top-2 margins there up to {(top2[1] - top2[0])[diff].max():.2e}, probability err {err:.2e}")
    assert int(diff.sum()) == 0, int(diff.sum())          # ONE mask bar across the suite: bit-exact (tests/test_configs_gpu.py)
    assert (out[cnt == 0] == 0).all()                              # nothing placed: class 0, as np.argmax of zeros


@pytest.mark.parametrize("case", ["u16", "bf16", "regression", "nonsquare"])
def test_raster_kernels_equal_the_tile_path(case):
    import predict as P
    from unet_amd import ops
    size = 128
    act = "bf16" if case == "bf16" else "f32"
    n_out = 1 if case == "regression" else 3
    model, _ = _pair("xresnet18", 3, n_out, size, seed=21, act_dtype=act)
    if case == "u16":
        img, dt = _raster(6, 3, 300, 290, np.uint16, hi=60000), "int16"
    else:
        img, dt = _raster(7, 3, 300, 350 if case == "nonsquare" else 290), "int8"
    kw = dict(regression=True) if case == "regression" else dict(all_classes=True)
    got = P.predict_raster(model, img, size, 0.25, dtype=dt, batch_size=4, **kw)
    zimg, wins = _reference_windows(img, size, 0.25, 0.9, None)
    mosaic, count = _tile_path_merge(model, zimg, wins, size, dt, n_out, regression=case == "regression", rep=4)
    ops.mosaic_finalize(mosaic, count, None)
    ref = mosaic.cpu().numpy()
    assert np.array_equal(got, ref[0] if case == "regression" else ref)
    # the same from a device tensor, and the argmax band alone
    if case != "regression":
        am = P.predict_raster(model, torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img).cuda().view(
            torch.uint16 if img.dtype == np.uint16 else torch.uint8), size, 0.25, dtype=dt, batch_size=4)
        assert np.array_equal(am, ref.argmax(0).astype(np.uint8))


@pytest.mark.parametrize("act_dtype", ["f32", "bf16"])
def test_predict_raster_batch_invariant_is_the_same_for_every_batch_size(act_dtype):
    """predict_raster(batch_invariant=True) = unet_tuning.plan_batch 1: every launch is planned as if its batch were one window, so windows in
    batches of 1, 3 and 7 (padded last batches included) give the SAME probabilities and the same mask bit for bit -- the reference predicts
    tile by tile (predict.py:191-193).  Without the flag the results agree to rounding level only (the planner follows the batch)."""
    import predict as P
    model, _ = _pair("xresnet18", 4, 3, 128, seed=23, act_dtype=act_dtype)
    img = _raster(9, 4, 300, 420)
    ref_mask = ref_probs = None
    for bs in (1, 3, 7):
        mask = P.predict_raster(model, img, 128, 0.2, batch_size=bs, batch_invariant=True)
        probs = P.predict_raster(model, img, 128, 0.2, batch_size=bs, all_classes=True, batch_invariant=True)
        if ref_mask is None:
            ref_mask, ref_probs = mask, probs
        else:
            assert np.array_equal(mask, ref_mask), bs
            assert np.array_equal(probs, ref_probs), bs
    loose = P.predict_raster(model, img, 128, 0.2, batch_size=7, all_classes=True)
    assert np.abs(loose - ref_probs).max() <= (2e-2 if act_dtype == "bf16" else 1e-4)


def test_window_kernels_unit():
    """nodata zeroing, non-zero counts and the window gather against numpy, every sample type"""
    from unet_amd import ops
    from unet_amd.learner import scale_input
    g = np.random.default_rng(3)
    for npdt, hi in ((np.uint8, 256), (np.uint16, 65536), (np.int16, 30000), (np.int32, 100000), (np.float32, 300)):
        a = (g.random((3, 90, 70)) * hi).astype(npdt) if npdt == np.float32 else g.integers(0, hi, (3, 90, 70)).astype(npdt)
        a[:, 10:30, 5:60] = 0
        a[2, 50, 7] = 77
        t = torch.from_numpy(a.view(np.int16)).view(torch.uint16) if npdt == np.uint16 else torch.from_numpy(a)
        src = ops.WindowSource(t.cuda().clone(), div255_twice=npdt == np.uint16)
        ops.raster_nodata_zero(src, 77.0)
        z = a.copy()
        z[:, (a == 77).any(axis=0)] = 0
        back = src.data.cpu()
        back = back.view(torch.int16).numpy().view(np.uint16) if npdt == np.uint16 else back.numpy()
        assert np.array_equal(back, z)
        wins = [(0, 0), (26, 6), (5, 38), (26, 6)]
        tab = ops.window_table([[y, x, 0, 0] for y, x in wins], "cuda")
        nz = ops.window_nonzero(src, tab, 64, 32).cpu().numpy()
        assert nz.tolist() == [int(np.sum(z[:, y:y + 64, x:x + 32] != 0)) for y, x in wins]
        for adt in (torch.float32, torch.bfloat16):
            cs = 8 if adt == torch.bfloat16 else 4
            buf = torch.full((4, 64, 32, cs), 7.0, dtype=adt, device="cuda")
            ops.window_gather(src, tab, 1, 3, 64, 32, buf, 0)
            for j, (y, x) in enumerate(wins[1:]):
                want = torch.from_numpy(scale_input(z[:, y:y + 64, x:x + 32], "int16" if npdt == np.uint16 else "int8")).permute(1, 2, 0).to(adt)
                assert torch.equal(buf[j, :, :, :3].cpu(), want), (npdt, adt, j)
            assert bool((buf[:3, :, :, 3:] == 7.0).all()) and bool((buf[3] == 7.0).all())       # pad lanes and unused slots untouched


def test_cfg5_full_size_properties():
    """configs[4] at its own size on one GPU: 20000 x 20000 4-band raster, 2401 windows of 512 at overlap 0.2, xresnet34 4 -> 5"""
    import predict as P
    from unet_amd import ops
    from unet_amd.model import HipDynamicUnet
    from unet_amd.mosaic import sliding_windows
    side, size = 20000, 512
    torch.manual_seed(0)
    model = HipDynamicUnet("xresnet34", 4, 5, (size, size))
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(3)
    raster = torch.randint(1, 256, (4, side, side), dtype=torch.uint8, device="cuda", generator=g)
    tm = {}
    out = P.predict_raster(model, raster, size, 0.2, batch_size=16, timing=tm)
    print(f"cfg5 fp32: {tm['seconds']:.2f} s = {tm['windows'] / tm['seconds']:.0f} tiles/s")
    assert out.shape == (side, side) and out.dtype == np.uint8 and tm["kept_windows"] == tm["windows"] == 2401
    assert tm["hits_min"] == 1 and tm["hits_max"] == 4                    # every pixel covered; corners of the overlap grid by four windows
    wins = sliding_windows(side, side, size, 0.2)
    ys = sorted(set(wins[:, 0].tolist()))
    # pixels covered by exactly ONE window carry that window's own argmax (mean of one probability vector = itself)
    for iy, ix in ((0, 0), (17, 30), (47, 47)):
        y, x = ys[iy], ys[ix]
        t = (raster[:, y:y + size, x:x + size].float() / 255.0)[None].repeat(16, 1, 1, 1)      # the engine's batch geometry
        _, am = model.predict_probs(t)
        lo_y = 0 if iy == 0 else ys[iy - 1] + size - y
        lo_x = 0 if ix == 0 else ys[ix - 1] + size - x
        hi_y, hi_x = ys[iy + 1] - y, ys[ix + 1] - x
        assert np.array_equal(am[0, lo_y:hi_y, lo_x:hi_x].cpu().numpy().astype(np.uint8), out[y + lo_y:y + hi_y, x + lo_x:x + hi_x])
    # a corner shared by four windows (incl. the ragged last row / column): merged by hand from the per-tile path
    for iy, ix in ((10, 20), (47, 47)):
        block = [(ys[iy + a], ys[ix + b]) for a in (0, 1) for b in (0, 1)]
        y1, x1 = block[3]
        h, w = block[0][0] + size - y1, block[0][1] + size - x1                      # the region all four cover
        mosaic = torch.zeros((5, h, w), dtype=torch.float32, device="cuda")
        count = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        for y, x in block:
            p, _ = model.predict_probs((raster[:, y:y + size, x:x + size].float() / 255.0)[None].repeat(16, 1, 1, 1))
            ops.mosaic_accumulate(p[0, :, y1 - y:y1 - y + h, x1 - x:x1 - x + w].contiguous(), mosaic, count, 0, 0)
        am = torch.empty((h, w), dtype=torch.uint8, device="cuda")
        ops.mosaic_finalize(mosaic, count, am)
        assert int(count.min()) == 4 and np.array_equal(am.cpu().numpy(), out[y1:y1 + h, x1:x1 + w])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _raster_worker(rank, world, port, state, img_path, outdir, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UNET_DIST_BACKEND="gloo", UNET_FORCE_DEVICE="0")
    import torch.distributed as dist
    import predict as P
    from unet_amd.model import HipDynamicUnet
    model = HipDynamicUnet("xresnet18", 4, 3, (512, 512), device="cuda:0")
    model.load_state_dict(torch.load(state))
    model.eval()
    img = np.load(img_path)
    tm = {}
    a = P.predict_raster(model, img, 512, 0.2, batch_size=4, timing=tm)
    b = P.predict_raster(model, img, 512, 0.2, batch_size=4, all_classes=True)
    if rank == 0:
        np.save(os.path.join(outdir, "a.npy"), a)
        np.save(os.path.join(outdir, "b.npy"), b)
    else:
        assert a is None and b is None
    q.put((rank, tm["windows_this_rank"], tm["active_ranks"], tm["strip_rows"], tm["slab_floats_sent"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_row_blocks_equal_one_rank(tmp_path):
    import predict as P
    model, _ = _pair("xresnet18", 4, 3, 512, seed=31)
    img = _raster(9, 4, 1500, 1300)
    one = P.predict_raster(model, img, 512, 0.2, batch_size=4)
    one_all = P.predict_raster(model, img, 512, 0.2, batch_size=4, all_classes=True)
    state = tmp_path / "w.pt"
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, state)
    np.save(tmp_path / "img.npy", img)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_raster_worker, args=(r, 2, port, str(state), str(tmp_path / "img.npy"), str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    # 12 windows (4 rows x 3): 6 per rank; rank 1 starts at window row 2 (y = 820) and owns rows from 410 + 512 = 922: its first three
    # windows send their top 102 rows to rank 0
    assert [r[:3] for r in res] == [(0, 6, 2), (1, 6, 2)], res
    assert res[0][3] == 922 and res[1][3] == 1500 - 922 and res[0][4] == 0 and res[1][4] == 3 * 3 * 102 * 512
    assert np.array_equal(np.load(tmp_path / "a.npy"), one) and np.array_equal(np.load(tmp_path / "b.npy"), one_all)
