"""Tile-DDP plumbing on ONE GPU: two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one
device).  Checks the backward hooks / bucket order / finish() on real HIP backward passes: after every step both ranks hold
identical parameters, and the reduced gradient equals the mean of the two ranks' local gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from unet_amd.distributed import broadcast_parameters, init_from_env
    from unet_amd.model import HipDynamicUnet
    from unet_amd.optimizer import FlatAdam
    from unet_amd.trainer import TrainStep
    init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                     # deliberately different initial weights per rank
    model = HipDynamicUnet("xresnet18", 4, 5, (64, 64), device="cuda:0")
    broadcast_parameters(model.flat_param, list(model.buffers()))
    model.mark_weights_dirty()
    model.train()
    g = torch.Generator().manual_seed(7 + rank)       # every rank draws its own tiles
    x = (torch.randint(0, 256, (2, 4, 64, 64), generator=g).float() / 255).cuda()
    y = torch.randint(0, 5, (2, 64, 64), generator=g).cuda()
    # local gradient of this rank (no reducer), pre-scaled by 1/world like the DDP step
    model.forward_loss_backward(x, y, None, grad_scale=1.0 / world)
    local = model.flat_grad.clone()
    # BN running stats were touched by that forward: restore identical state before the real steps
    broadcast_parameters(model.flat_param, list(model.buffers()))
    opt = FlatAdam(model, [1e-4, 3e-4, 1e-3])
    step = TrainStep(model, opt, None, world, max_bucket_elems=1 << 20)    # several buckets per span
    assert len(step.reducer.spans) > 8
    step.reducer.reset()
    seen = []
    inner = model.grad_ready_hook
    model.grad_ready_hook = lambda off: (seen.append((off, step.reducer._next)), inner(off))[1]
    loss = model.forward_loss_backward(x, y, None, grad_scale=1.0 / world)
    model.grad_ready_hook = inner
    # readiness points: final ResBlock + head first, then one per UnetBlock, the encoder children last -- strictly descending offsets,
    # and buckets already in flight after the FIRST point (the all-reduce starts while > 90 % of the backward is still to run)
    offs = [o for o, _ in seen]
    assert offs == sorted(offs, reverse=True) and len(set(offs)) == len(offs) and len(offs) >= 5 + 7, offs
    assert offs[0] == model._layer_offset[4 + len(model.sz_chg_idxs)] and seen[1][1] >= 1, seen[:3]
    step.reducer.finish()
    torch.cuda.synchronize()
    summed = local.clone()
    dist.all_reduce(summed)
    ok_grad = bool(((model.flat_grad - summed).abs().max() <= 1e-6 * summed.abs().max() + 1e-9).item())
    for _ in range(2):
        step(x, y)
    torch.cuda.synchronize()
    p = model.flat_param.clone()
    ref = p.clone()
    dist.broadcast(ref, 0)
    ok_param = bool(torch.equal(p, ref))
    q.put((rank, ok_grad, ok_param, float(loss.item())))
    dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert [r[:3] for r in res] == [(0, True, True), (1, True, True)], res


def _predict_worker(rank, world, port, pkl, tiles_dir, q, large_file=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UNET_DIST_BACKEND="gloo", UNET_FORCE_DEVICE="0")
    import torch.distributed as dist
    import predict as P
    out = P.save_predictions(pkl, tiles_dir, False, merge=True, all_classes=large_file, large_file=large_file, AOI="ddp8" if large_file else "ddp",
                             year=None, validation_vision=False, batch_size=2)
    q.put((rank, None if out is None else str(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharded_prediction_two_ranks(tmp_path):
    """BASELINE configs[4] partitioning (tile i -> rank i mod world, partial sum-of-probabilities mosaics all-reduced): the merged
    mask written by rank 0 of a 2-rank run equals the single-process result bit for bit"""
    import numpy as np
    import predict as P
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    from unet_amd.tiffio import read_tiff, write_tiff
    torch.manual_seed(3)
    model = HipDynamicUnet("xresnet18", 4, 3, (64, 64), device="cuda:0")
    dls = DataLoaders(TileDataset([np.zeros((4, 64, 64), np.uint8)], None, "int8"), None, 1, device="cuda:0", vocab=["a", "b", "c"])
    learn = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), metrics=[DiceMulti()], path=tmp_path)
    pkl = tmp_path / "m.pkl"
    learn.export(pkl)
    tiles = tmp_path / "set" / "tiles"
    tiles.mkdir(parents=True)
    g = np.random.default_rng(4)
    for i in range(5):                                   # 5 tiles, 50 % overlap along x: ranks get 3 + 2 tiles
        write_tiff(tiles / f"p{i}.tif", g.integers(0, 255, (4, 64, 64)).astype(np.uint8),
                   geotransform=(100.0 + i * 32 * 0.5, 0.5, 0.0, 900.0, 0.0, -0.5))
    single = P.save_predictions(pkl, tiles, False, merge=True, AOI="single", year=None, validation_vision=False, batch_size=2)
    ref, _ = read_tiff(single)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_worker, args=(r, 2, port, str(pkl), str(tiles), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res[1] is None and res[0] is not None
    got, meta = read_tiff(res[0])
    assert got.shape == ref.shape == (64, 192) and np.array_equal(got, ref)
    # large_file = True (reference predict.py:288-289,324-329): int8 rasters of around(p * 31), int8 hit counters, integer floor
    # division -- the 2-rank run (partial int8 rasters summed in int32 and wrapped back) equals the single process, which equals
    # the reference's host arithmetic restated in numpy
    single8 = P.save_predictions(pkl, tiles, False, merge=True, all_classes=True, large_file=True, AOI="single8", year=None,
                                 validation_vision=False, batch_size=2)
    ref8, _ = read_tiff(single8)
    lr = P.load_learner(pkl)
    acc = np.zeros((3, 64, 192), dtype=np.int8); cnt = np.zeros((3, 64, 192), dtype=np.int8)
    for i in range(5):
        _, _, pr = lr.predict(tiles / f"p{i}.tif")
        acc[:, :, i * 32:i * 32 + 64] += np.around(pr.numpy() * 31).astype(np.int8)
        cnt[:, :, i * 32:i * 32 + 64] += 1
    acc[cnt > 0] //= cnt[cnt > 0]
    assert ref8.dtype == np.int8 and np.array_equal(ref8, acc)
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_worker, args=(r, 2, port, str(pkl), str(tiles), q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    got8, _ = read_tiff(res[0])
    assert np.array_equal(got8, ref8)


def _rccl_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from unet_amd.distributed import GradReducer
    from unet_amd.model import HipDynamicUnet
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    model = HipDynamicUnet("xresnet18", 4, 5, (64, 64), device="cuda:0")
    model.train()
    g = torch.Generator().manual_seed(3)
    x = (torch.randint(0, 256, (2, 4, 64, 64), generator=g).float() / 255).cuda()
    y = torch.randint(0, 5, (2, 64, 64), generator=g).cuda()
    w = torch.tensor([0.1, 0.2, 0.3, 0.2, 0.2], device="cuda")
    model.forward_loss_backward(x, y, w)
    ref_grad, ref_loss = model.flat_grad.clone(), float(model.ctx.vec(model, "loss", 1).item())
    # the tile-DDP step's collective calls on RCCL itself: bucketed async all-reduce from the backward hooks, the loss
    # numerator / denominator exchange (world = 1 here, forced through the same code), broadcast, barrier, object gather
    bounds = [model._decoder_offset] + list(model._enc_child_offset.values())
    red = GradReducer(model.flat_grad, bounds, 1 << 20)
    red.world = 2                      # force the collective path with ONE rank: SUM over a world of 1 is the identity
    model.grad_ready_hook = red.ready_down_to
    red.reset()
    loss = model.forward_loss_backward(x, y, w, world=2)
    red.finish()
    torch.cuda.synchronize()
    ok = bool(torch.equal(model.flat_grad, ref_grad)) and abs(float(loss.item()) - ref_loss) <= 1e-6 * abs(ref_loss)
    dist.broadcast(model.flat_param, 0)
    for b in model.buffers():
        dist.broadcast(b, 0)
    dist.barrier()
    out = [None]
    dist.all_gather_object(out, {"rank": 0})
    q.put((ok, len(red.spans) > 4, dist.get_backend(), out[0]["rank"]))
    dist.destroy_process_group()


def test_rccl_api_path_with_one_rank():
    """RCCL itself (torch.distributed backend "nccl") on the one GPU of the box: the collective calls of the tile-DDP step --
    async bucketed all-reduce launched from the backward hooks on RCCL's stream, the 2-float loss exchange between the loss
    kernels, broadcast, barrier, all_gather_object -- run through the real library with a world of one rank, where every SUM is
    the identity: gradient and loss must be bit-identical to the plain step.  (The multi-rank arithmetic is covered over gloo.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res == (True, True, "nccl", 0), res
