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
    loss = model.forward_loss_backward(x, y, None, grad_scale=1.0 / world)
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
