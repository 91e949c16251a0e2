"""Learner-surface pieces on the device: DiceMulti and the flip augmentation against the oracle / the reference's slicing rule,
fastai-layout model files, and the tile-DDP loss exchange (two ranks sharing one GPU over gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_dice_multi_on_device_matches_oracle():
    """fastai DiceMulti(axis=1) (reference train.py:196): counters accumulated on the GPU over several validation batches,
    classes absent from a batch, a class absent from the whole set (nan skipped)"""
    from unet_amd.learner import DiceMulti
    g = torch.Generator().manual_seed(0)
    a, b = DiceMulti(), O.DiceMulti()
    for i in range(4):
        logits = torch.randn(3, 6, 40, 56, generator=g)
        logits[:, 5] = -50.0                                   # class 5 is never predicted ...
        targ = torch.randint(0, 5 if i else 3, (3, 40, 56), generator=g)   # ... and never present
        a.accumulate_argmax(logits.cuda().argmax(1), targ.cuda(), 6)
        b.accumulate(logits, targ)
    assert a.inter.is_cuda and abs(a.value - b.value) < 1e-12


def test_flip_augment_on_device_follows_the_reference_slicing_rule():
    """utils.py:239-295: only the first n_transform - B images of a batch are candidates (python slice semantics), each flipped
    horizontally / vertically with p = 0.5 (params_and_main.py:105-108); image and mask move together; n_transform_imgs = 1
    (the shipped default) touches nothing (quirk Q7)"""
    from unet_amd.learner import FlipAugment
    g = torch.Generator().manual_seed(1)
    x = torch.rand(8, 4, 16, 12, generator=g).cuda()
    y = torch.randint(0, 3, (8, 16, 12), generator=g).cuda()
    x0, y0 = x.clone(), y.clone()
    xa, ya = FlipAugment(n_transform_imgs=1.0)(x.clone(), y.clone())
    assert torch.equal(xa, x0) and torch.equal(ya, y0)
    aug = FlipAugment(n_transform_imgs=0.5, seed=3)
    xa, ya = aug(x.clone(), y.clone())
    assert xa.is_cuda
    # ceil(8 * 0.5) - 8 = -4: images 0..3 are candidates, 4..7 never change
    assert torch.equal(xa[4:], x0[4:]) and torch.equal(ya[4:], y0[4:])
    seen = set()
    for i in range(4):
        for fh in (False, True):
            for fv in (False, True):
                xe, ye = x0[i], y0[i]
                if fh:
                    xe, ye = xe.flip(-1), ye.flip(-1)
                if fv:
                    xe, ye = xe.flip(-2), ye.flip(-2)
                if torch.equal(xa[i], xe) and torch.equal(ya[i], ye):
                    seen.add((i, fh, fv))
    assert {s[0] for s in seen} == {0, 1, 2, 3}          # every candidate is one of the four flip states, image and mask alike
    # statistics of the flip decisions over many draws: p_h = p_v = 0.5
    aug = FlipAugment(n_transform_imgs=0.01, seed=5)     # ceil(0.08) - 8 = -7: the first image only
    nh = nv = 0
    for _ in range(400):
        xa, _ = aug(x0.clone(), y0.clone())
        nh += int(torch.equal(xa[0], x0[0].flip(-1)) or torch.equal(xa[0], x0[0].flip(-1).flip(-2)))
        nv += int(torch.equal(xa[0], x0[0].flip(-2)) or torch.equal(xa[0], x0[0].flip(-1).flip(-2)))
    assert 150 < nh < 250 and 150 < nv < 250


def _tiny_learner(tmp_path, n_cls=3):
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, DiceMulti, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    g = np.random.default_rng(0)
    imgs = [g.integers(0, 255, (4, 64, 64)).astype(np.uint8) for _ in range(4)]
    masks = [g.integers(0, n_cls, (64, 64)).astype(np.uint8) for _ in range(4)]
    model = HipDynamicUnet("xresnet18", 4, n_cls, (64, 64))
    dls = DataLoaders(TileDataset(imgs, masks, "int8"), TileDataset(imgs[:2], masks[:2], "int8"), 2, vocab=list("abc"))
    return Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), metrics=[DiceMulti()], path=tmp_path)


def test_model_files_use_fastai_layouts(tmp_path):
    """fastai save_model / load_model: with_opt=False (SaveModelCallback) writes the BARE state_dict, with_opt=True writes
    {'model','opt'}; load accepts both, and a foreign optimizer state (as written by fastai's own Adam) is skipped with a warning"""
    learn = _tiny_learner(tmp_path)
    learn.create_opt()
    learn.save("bare")
    sd = torch.load(tmp_path / "models" / "bare.pth")
    assert "model" not in sd and "layers.0.0.0.weight" in sd
    learn.save("full", with_opt=True)
    sd2 = torch.load(tmp_path / "models" / "full.pth")
    assert set(sd2) == {"model", "opt"}
    def params():
        return torch.cat([p.detach().flatten() for p in learn.model.parameters()]).clone()

    def disturb():
        with torch.no_grad():
            for p in learn.model.parameters():
                p.add_(1.0)
    w0 = params()
    disturb()
    learn.load("bare")
    assert torch.equal(params(), w0)
    disturb()
    learn.load("full", with_opt=True)
    assert torch.equal(params(), w0)
    # a fastai-written {'model','opt'} file: opt state = {'hypers': [...], 'state': [{'grad_avg':..,'sqr_avg':..,'step':..}, ...]}
    torch.save({"model": sd, "opt": {"hypers": [{"lr": 1e-3}], "state": [{"grad_avg": torch.zeros(3), "step": 1}]}},
               tmp_path / "models" / "fastai.pth")
    disturb()
    with pytest.warns(UserWarning, match="optimizer state"):
        learn.load("fastai", with_opt=True)
    assert torch.equal(params(), w0)
    with pytest.warns(UserWarning, match="doesn't contain an optimizer state"):
        learn.load("bare", with_opt=True)


def _ce_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from unet_amd.distributed import broadcast_parameters, init_from_env
    from unet_amd.learner import DiceMulti
    from unet_amd.model import HipDynamicUnet
    init_from_env(backend="gloo")
    torch.manual_seed(50)
    model = HipDynamicUnet("xresnet18", 4, 5, (64, 64), device="cuda:0")
    broadcast_parameters(model.flat_param, list(model.buffers()))
    model.mark_weights_dirty()
    model.train()
    g = torch.Generator().manual_seed(9 + rank)
    x = (torch.randint(0, 256, (2, 4, 64, 64), generator=g).float() / 255).cuda()
    # rank 0 sees mostly class 0, rank 1 mostly class 4: with non-uniform weights the ranks' denominators differ a lot
    y = torch.where(torch.rand(2, 64, 64, generator=g) < 0.8, torch.full((2, 64, 64), 4 * rank), torch.randint(0, 5, (2, 64, 64), generator=g)).cuda()
    w = torch.tensor([0.02, 0.1, 0.2, 0.3, 1.5], device="cuda")
    loss = model.forward_loss_backward(x, y, w, world=world)
    torch.cuda.synchronize()
    z = model.logits_ts().view().permute(0, 3, 1, 2).contiguous()
    dz = model.ctx.act(model, "dlogits", 2, 64, 64, 5, zero=True).view().permute(0, 3, 1, 2).contiguous()
    zs, ys = [torch.empty_like(z) for _ in range(world)], [torch.empty_like(y) for _ in range(world)]
    dist.all_gather(zs, z); dist.all_gather(ys, y)
    # oracle: ONE weighted cross-entropy over the logits / targets of all ranks
    zc = torch.cat(zs).cpu().double().requires_grad_(True)
    l_ref = O.CrossEntropyLossFlat(weight=w.cpu().double())(zc, torch.cat(ys).cpu())
    l_ref.backward()
    dz_ref = zc.grad[2 * rank:2 * rank + 2]
    ok_loss = abs(loss.item() - l_ref.item()) <= 2e-6 * abs(l_ref.item())
    ok_dz = (dz.cpu().double() - dz_ref).abs().max().item() <= 2e-6 * dz_ref.abs().max().item()
    # DiceMulti counters are summed over the ranks: every rank reports the value of the whole validation set
    dm = DiceMulti()
    dm.accumulate_argmax(z.argmax(1), y, 5)
    dm.all_reduce()
    ref = O.DiceMulti()
    ref.accumulate(torch.cat(zs).cpu(), torch.cat(ys).cpu())
    ok_dice = abs(dm.value - ref.value) < 1e-12
    # a rank whose tiles carry only zero-weight classes has denominator 0 -- the ranks exchange numerator and denominator
    # themselves (unet_ce_fwd_parts), so the global loss stays the finite single-process value instead of turning NaN through a 0/0
    w0 = torch.tensor([0.0, 0.1, 0.2, 0.3, 1.5], device="cuda")
    y0 = torch.zeros_like(y) if rank == 1 else y
    loss0 = model.forward_loss_backward(x, y0, w0, world=world)
    z0 = model.logits_ts().view().permute(0, 3, 1, 2).contiguous()
    zs0, ys0 = [torch.empty_like(z0) for _ in range(world)], [torch.empty_like(y0) for _ in range(world)]
    dist.all_gather(zs0, z0); dist.all_gather(ys0, y0)
    l0 = O.CrossEntropyLossFlat(weight=w0.cpu().double())(torch.cat(zs0).cpu().double(), torch.cat(ys0).cpu())
    ok_zero = bool(torch.isfinite(loss0).all().item()) and abs(loss0.item() - l0.item()) <= 2e-6 * abs(l0.item())
    ok_zero = ok_zero and bool(torch.isfinite(model.flat_grad).all().item())
    # FocalLossFlat over the ranks (params_and_main.py:87-89): a plain mean over equally many pixels per rank -- loss averaged over the ranks,
    # every rank's logit gradient that of ONE focal loss over the global batch
    lossf = model.forward_loss_backward(x, y, w, world=world, focal_gamma=2.0)
    torch.cuda.synchronize()
    zf = model.logits_ts().view().permute(0, 3, 1, 2).contiguous()
    dzf = model.ctx.act(model, "dlogits", 2, 64, 64, 5, zero=True).view().permute(0, 3, 1, 2).contiguous()
    zsf = [torch.empty_like(zf) for _ in range(world)]
    dist.all_gather(zsf, zf)
    zcf = torch.cat(zsf).cpu().double().requires_grad_(True)
    lf_ref = O.FocalLossFlat(gamma=2.0, weight=w.cpu().double())(zcf, torch.cat(ys).cpu())
    lf_ref.backward()
    ok_focal = abs(lossf.item() - lf_ref.item()) <= 2e-6 * abs(lf_ref.item())
    ok_focal = ok_focal and (dzf.cpu().double() - zcf.grad[2 * rank:2 * rank + 2]).abs().max().item() <= 2e-6 * zcf.grad.abs().max().item()
    q.put((rank, bool(ok_loss), bool(ok_dz), bool(ok_dice and ok_zero and ok_focal)))
    dist.destroy_process_group()


def test_weighted_ce_numerator_denominator_and_dice_counters_are_all_reduced():
    """with non-uniform class weights sum w[y] differs per rank; loss and logit-gradient of every rank must be those
    of ONE cross-entropy over the global batch (numerator and denominator all-reduced between the loss kernels)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(0, True, True, True), (1, True, True, True)], res


def _learner_worker(rank, world, port, root, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pathlib import Path
    from unet_amd.distributed import init_from_env
    from unet_amd.learner import CSVLogger, SaveModelCallback
    init_from_env(backend="gloo")
    torch.manual_seed(1000 + rank)                       # different initial weights per rank: the Learner must broadcast
    learn = _tiny_learner(Path(root))
    learn.cbs = [SaveModelCallback(monitor="valid_loss"), CSVLogger()]
    assert learn.world == 2 and learn.dls.train.world == 2 and len(learn.dls.train) == 1
    learn.fit_one_cycle(2, lr_max=slice(1e-4, 1e-3))
    p = learn.model.flat_param.clone()
    ref = p.clone()
    dist.broadcast(ref, 0)
    q.put((rank, bool(torch.equal(p, ref)), [float(v) for v in learn.recorder.values[-1][:3]]))
    dist.barrier()
    dist.destroy_process_group()


def test_learner_tile_ddp_two_ranks(tmp_path):
    """the Learner under an initialised process group: rank-sharded training loader, replicas broadcast from rank 0, one
    history.csv / best-model.pth (rank 0), validation loss and DiceMulti identical on both ranks (all-reduced)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res[0][1] and res[1][1], res
    assert res[0][2][1:] == res[1][2][1:], res           # valid_loss, dice_multi: global values on every rank
    hist = (tmp_path / "history.csv").read_text().strip().splitlines()
    assert hist[0] == "epoch,train_loss,valid_loss,dice_multi,time" and len(hist) == 3
    assert (tmp_path / "models" / "best-model.pth").exists()


def _empty_shard_worker(rank, world, port, root, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pathlib import Path
    from unet_amd.distributed import init_from_env
    from unet_amd.learner import DataLoaders, Rmse, TileDataset, load_learner
    init_from_env(backend="gloo")
    torch.manual_seed(7)                                  # identical replicas: the single-process value is the reference
    learn = _tiny_learner(Path(root))
    vd = learn.dls.valid_ds
    # ONE validation tile over two ranks: rank 1's shard is empty and must still take part in the (single) collective
    one = DataLoaders(learn.dls.train_ds, TileDataset(vd.imgs[:1], vd.masks[:1], "int8"), 2, vocab=list("abc"))
    learn.dls = one                                       # the setter shards: the fine-tune branch of train_unet assigns dls like this
    sharded = (learn.dls.train.world, learn.dls.valid.world, len(learn.dls.train), len(learn.dls.valid))
    v1 = learn.validate()
    v2 = learn.validate()                                 # a second pass pairs its collectives correctly too
    # fine-tune path (reference train.py:225-229): load_learner on an exported file, then learn.dls = dls
    if rank == 0:
        learn.export(Path(root) / "ft.pkl")
    dist.barrier()
    ft = load_learner(Path(root) / "ft.pkl", device="cuda:0")
    ft.dls = one
    q.put((rank, sharded, [float(v) for v in v1], [float(v) for v in v2], (ft.world, ft.dls.train.world, len(ft.dls.train))))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_with_an_empty_rank_shard_and_finetune_sharding(tmp_path):
    """(a) len(valid) < world leaves a rank without validation tiles -- metric counters and loss sums travel in ONE
    all-reduce that every rank joins, values equal the single-process ones; (b) `learn.dls = dls` after load_learner shards the loaders"""
    torch.manual_seed(7)
    single = _tiny_learner(tmp_path / "single")
    from unet_amd.learner import DataLoaders, TileDataset
    vd = single.dls.valid_ds
    single.dls = DataLoaders(single.dls.train_ds, TileDataset(vd.imgs[:1], vd.masks[:1], "int8"), 2, vocab=list("abc"))
    ref = [float(v) for v in single.validate()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] == (2, 2, 1, 1 if r[0] == 0 else 0), r            # 4 train tiles / 2 ranks / bs 2 = 1 step; the valid tile sits on rank 0
        assert r[2] == r[3] and np.allclose(r[2], ref, rtol=1e-6, atol=0), (r, ref)
        assert r[4] == (2, 2, 1), r


def test_augmentation_pipeline_runs_on_the_device():
    """the albumentations-semantics pipeline (unet_amd.augment) on a CUDA batch: values stay on the device, the batch rule of
    utils.py:239-291 holds, brightness / contrast stays in [0, 1], dropout holes hit every band and never the mask"""
    from unet_amd import augment as A
    g = torch.Generator().manual_seed(2)
    x = torch.rand(6, 4, 64, 64, generator=g).cuda()
    y = torch.randint(1, 3, (6, 64, 64), generator=g).cuda()
    x0, y0 = x.clone(), y.clone()
    pipe = A.Compose([A.HorizontalFlip(p=0.5), A.VerticalFlip(p=0.5),
                      A.RandomBrightnessContrast(brightness_limit=(-0.1, 0.1), contrast_limit=(-0.1, 0.1), p=1.0), A.CoarseDropout(p=1.0)])
    xa, ya = A.BatchAugment(pipe, n_transform_imgs=0.5, seed=1)(x, y)
    assert xa.is_cuda and torch.equal(xa[3:], x0[3:]) and torch.equal(ya[3:], y0[3:])
    assert float(xa.min()) >= 0.0 and float(xa.max()) <= 1.0
    for i in range(3):
        assert not torch.equal(xa[i], x0[i])
        holes = (xa[i] == 0).all(0)
        assert 64 <= int(holes.sum()) <= 512
        assert int((ya[i] == 0).sum()) == 0                      # mask_fill_value None: the mask keeps its classes
        assert sorted(ya[i].flatten().tolist()) == sorted(y0[i].flatten().tolist())      # a flip permutes the mask, nothing else


def test_graphed_steps_with_device_resident_inputs_use_their_own_hyper_parameters():
    """with inputs already on the device nothing blocks the host, which runs several graph replays ahead of the stream.
    The Adam hyper-parameters (lr, momentum, de-bias terms: different in EVERY step of a one-cycle schedule) are staged through a ring
    of pinned blocks guarded by events, so replay k must still read the values of step k: the graphed trajectory equals the eager one."""
    from unet_amd.model import HipDynamicUnet
    from unet_amd.optimizer import FlatAdam
    from unet_amd.trainer import TrainStep
    torch.manual_seed(13)
    sd = O.DynamicUnet("xresnet18", 4, 5, (64, 64)).state_dict()
    x, y = O.synthetic_batch(2, 4, 64, 64, 5)
    xd, yd = x.cuda(), y.cuda()
    finals = []
    for use_graph in (False, True):
        model = HipDynamicUnet("xresnet18", 4, 5, (64, 64))
        model.load_state_dict(sd)
        model.train()
        opt = FlatAdam(model, [1e-4, 3e-4, 1e-3])
        step = TrainStep(model, opt, None, 1, use_graph=use_graph)
        for i in range(40):                        # no host synchronisation inside this loop
            opt.set_lr([1e-4 * (1 + i % 7), 3e-4 * (1 + i % 5), 1e-3 / (1 + i % 3)])
            opt.mom = 0.95 - 0.002 * i
            step(xd, yd)
        torch.cuda.synchronize()
        finals.append(model.flat_param.clone())
    assert (finals[0] - finals[1]).abs().max().item() < 1e-6
