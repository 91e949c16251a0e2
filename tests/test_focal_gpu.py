"""FocalLossFlat(gamma, axis=1) -- the alternative classification loss the reference's configuration names (params_and_main.py:87-89) -- on
the device (unet_focal_fwd / unet_focal_bwd) against the oracle's restatement of fastai 2.5.1 ``FocalLoss`` (values and autograd gradients),
through the training step and through ``Learner.fit_one_cycle``."""
import copy

import numpy as np
import pytest
import torch

from util import empty_ts, from_ts, to_ts

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)


@pytest.mark.parametrize("gamma", [2.0, 0.5, 0.0, 1.0])
@pytest.mark.parametrize("weighted", [False, True])
def test_focal_kernels_against_the_oracle(gamma, weighted):
    """loss value and logit gradient per pixel, on a channel slice of a wider buffer, ragged pixel count; gamma = 0 is the (plain-mean)
    cross-entropy; fp64 autograd of the oracle is the reference for the gradient"""
    from unet_amd import ops
    g = torch.Generator().manual_seed(int(gamma * 10) + weighted)
    N, C, H, W = 3, 5, 37, 29
    z = torch.randn(N, C, H, W, generator=g) * 3.0
    y = torch.randint(0, C, (N, H, W), generator=g)
    w = (torch.rand(C, generator=g) + 0.3) if weighted else None
    z64 = z.double().requires_grad_(True)
    l64 = O.FocalLossFlat(gamma=gamma, weight=None if w is None else w.double())(z64, y)
    l64.backward()
    l32 = O.FocalLossFlat(gamma=gamma, weight=w)(z, y)
    zt = to_ts(z, cs=12, co=4)
    loss = torch.zeros(1, device="cuda")
    ws = torch.empty(ops.ce_workspace(zt.P), device="cuda")
    wd = None if w is None else w.cuda()
    ops.focal_fwd(zt, y.cuda().contiguous(), wd, gamma, loss, ws)
    assert abs(loss.item() - l64.item()) <= max(2e-6 * abs(l64.item()), 3 * abs(l32.item() - l64.item())), (loss.item(), l64.item(), l32.item())
    dz = empty_ts(N, H, W, C, cs=8, co=0)
    ops.focal_bwd(zt, y.cuda().contiguous(), wd, gamma, 0.5, dz)
    got = from_ts(dz).double()
    ref = 0.5 * z64.grad
    assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-12
    # bf16 gradient slice: one rounding of the fp32 value
    dzb = ops.TS(torch.zeros((N, H, W, 8), dtype=torch.bfloat16, device="cuda"), 0, C)
    ops.focal_bwd(zt, y.cuda().contiguous(), wd, gamma, 0.5, dzb)
    gb = dzb.view().permute(0, 3, 1, 2).float().cpu().double()
    assert (gb - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()


def test_focal_saturated_pixels_and_ignored_targets():
    """a pixel whose cross-entropy is exactly 0 in fp32 (logit margin beyond exp's range) contributes 0 loss and a 0 gradient for every gamma
    (torch's autograd gives NaN there for gamma < 1: 0 * inf); a target outside [0, C) contributes nothing but still counts in the mean"""
    from unet_amd import ops
    z = torch.zeros(1, 3, 2, 2)
    z[0, :, 0, 0] = torch.tensor([40.0, -40.0, -40.0])      # ce == 0 exactly
    z[0, :, 0, 1] = torch.tensor([0.3, -0.2, 0.1])
    z[0, :, 1, 0] = torch.tensor([1.0, 2.0, 3.0])
    y = torch.tensor([[[0, 2], [-100, 1]]])
    zt = to_ts(z)
    for gamma in (0.5, 2.0):
        loss = torch.zeros(1, device="cuda")
        ops.focal_fwd(zt, y.cuda(), None, gamma, loss, torch.empty(ops.ce_workspace(4), device="cuda"))
        dz = empty_ts(1, 2, 2, 3)
        ops.focal_bwd(zt, y.cuda(), None, gamma, 1.0, dz)
        got = from_ts(dz)
        assert torch.isfinite(got).all() and torch.isfinite(loss).all()
        assert got[0, :, 0, 0].abs().max().item() == 0.0 and got[0, :, 1, 0].abs().max().item() == 0.0
        keep = torch.tensor([[[False, True], [False, True]]])
        zz = z.double().requires_grad_(True)
        ce = torch.nn.functional.cross_entropy(zz, y.clamp(min=0), reduction="none")
        ref = (((1 - torch.exp(-ce)) ** gamma * ce) * keep).sum() / 4
        ref.backward()
        assert abs(loss.item() - ref.item()) < 1e-6
        assert (got.double() - zz.grad)[0, :, 0, 1].abs().max().item() < 1e-6 and (got.double() - zz.grad)[0, :, 1, 1].abs().max().item() < 1e-6


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_training_step_with_focal_loss(dtype):
    """forward + FocalLossFlat + backward of the whole network against the oracle network + oracle loss (smooth network: no ReLU flips);
    bf16 storage within its stated network tolerances"""
    from unet_amd.model import HipDynamicUnet
    import torch.nn as nn
    arch, n_in, n_out, size, bs = "xresnet18", 4, 3, (64, 64), 2
    torch.manual_seed(3)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=4)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.bias.fill_(8.0)
            elif isinstance(m, nn.Conv2d) and m.bias is not None:
                m.weight.mul_(0.01)
                m.bias.fill_(1.0)
    model = HipDynamicUnet(arch, n_in, n_out, size, act_dtype=dtype)
    model.load_state_dict(ref.state_dict())
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    w = torch.tensor([0.5, 1.5, 1.0])
    ref.train(); model.train()
    loss_ref = O.FocalLossFlat(gamma=2.0, weight=w)(ref(x), y)
    loss_ref.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda(), focal_gamma=2.0)
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == "f32" else 3e-2
    assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    g_hip = torch.cat([p.grad.flatten().cpu() for p in model.parameters()])
    g_ref = torch.cat([p.grad.flatten() for p in ref.parameters()])
    cos = torch.nn.functional.cosine_similarity(g_hip.double(), g_ref.double(), dim=0).item()
    assert cos > (1 - 1e-6 if dtype == "f32" else 0.99), cos
    if dtype == "f32":
        worst = max((p.grad.cpu() - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
                    for p, q in zip(model.parameters(), ref.parameters()) if q.grad.abs().max().item() > 1e-20)
        assert worst < 2e-3, worst


def test_learner_fits_validates_exports_with_focal_loss(tmp_path):
    """train.train_unet's sequence with loss_func=FocalLossFlat(gamma=2): class weights assigned to .func.weight (train.py:211), one epoch,
    validation loss = the oracle's focal loss of the validation set, export / load_learner keep the loss and its gamma"""
    from unet_amd.learner import DataLoaders, DiceMulti, FocalLossFlat, Learner, TileDataset, load_learner
    from unet_amd.model import HipDynamicUnet
    g = np.random.default_rng(0)
    imgs = [g.integers(0, 255, (4, 64, 64)).astype(np.uint8) for _ in range(4)]
    masks = [g.integers(0, 3, (64, 64)).astype(np.uint8) for _ in range(4)]
    torch.manual_seed(1)
    model = HipDynamicUnet("xresnet18", 4, 3, (64, 64))
    dls = DataLoaders(TileDataset(imgs, masks, "int8"), TileDataset(imgs[:3], masks[:3], "int8"), 2, vocab=list("abc"))
    loss = FocalLossFlat(gamma=2, axis=1)
    loss.func.weight = torch.tensor([0.2, 0.3, 0.5])
    learn = Learner(dls, model, loss_func=loss, metrics=[DiceMulti()], path=tmp_path)
    learn._no_logging = True
    learn.fit_one_cycle(1, lr_max=slice(1e-4, 1e-3))
    torch.cuda.synchronize()
    assert len(learn.recorder.losses) == 2 and all(np.isfinite(learn.recorder.losses))
    # validation loss against the oracle evaluated on the trained weights (eval mode: running statistics)
    ref = O.DynamicUnet("xresnet18", 4, 3, (64, 64))
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        xs = torch.from_numpy(np.stack(imgs[:3]).astype(np.float32) / 255.0)
        ys = torch.from_numpy(np.stack(masks[:3]).astype(np.int64))
        # Learner.validate sums loss * pixels per batch (2 + 1 tiles): the plain mean over all pixels of the set
        want = O.FocalLossFlat(gamma=2.0, weight=torch.tensor([0.2, 0.3, 0.5]))(ref(xs), ys).item()
    got = learn.validate()[0]
    assert abs(got - want) < 1e-4 * max(1.0, abs(want)), (got, want)
    learn.export(tmp_path / "focal.pkl")
    back = load_learner(tmp_path / "focal.pkl")
    assert isinstance(back.loss_func, FocalLossFlat) and back.loss_func.gamma == 2.0
    assert torch.allclose(torch.as_tensor(back.loss_func.func.weight), torch.tensor([0.2, 0.3, 0.5]))
