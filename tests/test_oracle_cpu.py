"""CPU suite (runs with -m "not gpu"): the oracle against its committed golden vectors, known-answer tests of the
restated fastai pieces, structural anchors from SURVEY.md section 8(c)."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"


def _rebuild(g):
    torch.manual_seed(int(g["seed"]))
    m = O.DynamicUnet(str(g["arch"]), int(g["n_in"]), int(g["n_out"]), tuple(int(v) for v in g["size"]))
    O.randomize_bn_and_zero_gammas(m, seed=int(g["seed"]) + 1)
    return m


@pytest.mark.parametrize("name", ["net_x34_4to5_64", "net_x18_3to2_80"])
def test_oracle_reproduces_golden(name):
    g = np.load(GOLD / f"{name}.npz")
    m = _rebuild(g)
    cs = float(sum(p.double().abs().sum() for p in m.parameters()))
    assert abs(cs - float(g["param_checksum"])) < 1e-6 * cs, "seeded initialisation drifted"
    x, y, w = torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), torch.from_numpy(g["w"])
    m.eval()
    with torch.no_grad():
        z = m(x)
    assert np.abs(z.numpy() - g["z_eval"]).max() < 1e-4
    assert np.array_equal(z.argmax(1).numpy().astype(np.uint8), g["argmax_eval"])
    m.train()
    zt = m(x)
    loss = O.CrossEntropyLossFlat(weight=w)(zt, y)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    assert np.abs(m.layers[12][0].bias.grad.numpy() - g["g_head_b"]).max() < 1e-5 * (1 + np.abs(g["g_head_b"]).max())


def test_structural_anchors():
    m = O.DynamicUnet("xresnet34", 4, 5, (64, 64))
    assert O.count_params(m) == 41_244_577                 # SURVEY.md 8(c) anchor
    assert m.sz_chg_idxs == [6, 5, 4, 2]
    keys = list(m.state_dict().keys())
    assert keys[0] == "layers.0.0.0.weight" and "layers.12.0.bias" in keys and "layers.11.convpath.1.0.weight" in keys
    assert [sum(p.numel() for p in g) for g in O.xresnet_split(m)] == [29056, 21275136, 19940385]
    assert O.count_params(O.DynamicUnet("xresnet18", 3, 2, (64, 64))) == 31_132_240
    # decoder convs carry a bias and no norm (the NormType-class quirk, train.py:100,142)
    assert m.layers[4].conv1[0].bias is not None and len(m.layers[4].conv1) == 2
    # output size == input size also for tiles not divisible by 32 (reference default patch_size 400)
    with torch.no_grad():
        assert O.DynamicUnet("xresnet18", 3, 2, (80, 80)).eval()(torch.rand(1, 3, 80, 80)).shape == (1, 2, 80, 80)


def test_cross_entropy_flat_known_answer():
    """CrossEntropyLossFlat(axis=1, weight) == sum w[y] nll / sum w[y]; both operands are transposed, so a
    non-square tile still pairs pixel i with target i."""
    g = torch.Generator().manual_seed(0)
    z = torch.randn(2, 4, 5, 7, generator=g)
    y = torch.randint(0, 4, (2, 5, 7), generator=g)
    w = torch.tensor([0.5, 1.0, 2.0, 4.0])
    got = O.CrossEntropyLossFlat(weight=w)(z, y)
    lp = F.log_softmax(z, dim=1)
    nll = -lp.gather(1, y[:, None]).squeeze(1)
    assert abs(float(got) - float((w[y] * nll).sum() / w[y].sum())) < 1e-6
    lf = O.CrossEntropyLossFlat()
    assert torch.equal(lf.decodes(z), z.argmax(1)) and torch.allclose(lf.activation(z).sum(1), torch.ones(2, 5, 7))


def test_dice_multi_known_answer():
    d = O.DiceMulti()
    pred = torch.zeros(1, 3, 2, 2)
    pred[0, 0, 0, 0] = pred[0, 1, 0, 1] = pred[0, 1, 1, 0] = pred[0, 2, 1, 1] = 5.0   # argmax: [[0,1],[1,2]]
    targ = torch.tensor([[[0, 1], [2, 2]]])
    d.accumulate(pred, targ)
    # class0: inter 1 union 2 -> 1.0 ; class1: inter 1 union 3 -> 2/3 ; class2: inter 1 union 3 -> 2/3
    assert abs(d.value - (1.0 + 2 / 3 + 2 / 3) / 3) < 1e-9
    d.reset()
    d.accumulate(torch.tensor([[[[1.0]], [[0.0]]]]), torch.tensor([[[0]]]))
    assert abs(d.value - 1.0) < 1e-9        # class 1 never appears: nan is skipped by nanmean


def test_fastai_adam_and_one_cycle_golden():
    g = np.load(GOLD / "optim.npz")
    ps = [torch.nn.Parameter(torch.from_numpy(v).clone()) for v in g["p0"]]
    lr_f, mom_f = O.one_cycle_scheds(O.even_mults(1e-3 / 10, 1e-3, 3))
    opt = O.FastaiAdam([[p] for p in ps], lr_f(0.0), no_wd=[ps[1]])
    for it, pct in enumerate((0.0, 0.5)):
        opt.lrs, opt.mom = list(lr_f(pct)), float(mom_f(pct))
        for p, gg in zip(ps, g["grads"][it]):
            p.grad = torch.from_numpy(gg).clone()
        opt.step()
    assert np.abs(torch.stack([p.detach() for p in ps]).numpy() - g["p2"]).max() < 1e-7
    for t, lr, mom in zip(g["pcts"], g["lr"], g["mom"]):
        assert np.allclose(lr_f(float(t)), lr) and abs(mom_f(float(t)) - mom) < 1e-12
    # hand-checked schedule anchors: lr_max/25 at 0, lr_max at pct_start, lr_max/1e5 at the end; moms 0.95/0.85/0.95
    lr_max = O.even_mults(1e-4, 1e-3, 3)
    assert np.allclose(lr_max, [1e-4, 1e-4 * math.sqrt(10), 1e-3])
    assert np.allclose(lr_f(0.0), lr_max / 25) and np.allclose(lr_f(0.25), lr_max) and np.allclose(lr_f(1.0), lr_max / 1e5)
    assert abs(mom_f(0.0) - 0.95) < 1e-12 and abs(mom_f(0.25) - 0.85) < 1e-12 and abs(mom_f(1.0) - 0.95) < 1e-12


def test_adam_first_step_by_hand():
    """t=1: m = (1-mom) g, v = (1-sqr) g^2, debias -> p -= lr * g / (|g| + eps); decoupled wd first."""
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
    p.grad = torch.tensor([0.5, -0.25])
    O.FastaiAdam([[p]], 0.1, wd=0.01).step()
    exp = torch.tensor([1.0, -2.0]) * (1 - 0.1 * 0.01) - 0.1 * p.grad / (p.grad.abs() + 1e-5)
    assert torch.allclose(p.detach(), exp, atol=1e-6)


def test_blur_and_shuffle_semantics():
    """PixelShuffle(2): out[c,2h+i,2w+j] = in[4c+2i+j,h,w]; blur = replicate-pad (left, top) + 2x2 mean."""
    x = torch.arange(16.0).view(1, 4, 2, 2)
    ps = F.pixel_shuffle(x, 2)
    assert ps[0, 0, 1, 0] == x[0, 2, 0, 0] and ps[0, 0, 0, 1] == x[0, 1, 0, 0]
    blk = O.PixelShuffleICNR(8, 2, blur=True)
    u = torch.rand(1, 2, 4, 4)
    b = blk[3](blk[2](u))
    assert b.shape == u.shape
    assert abs(float(b[0, 0, 0, 0]) - float(u[0, 0, 0, 0])) < 1e-6
    assert abs(float(b[0, 0, 2, 1]) - float(u[0, 0, 1:3, 0:2].mean())) < 1e-6


def test_merge_golden_fixture_is_self_consistent():
    g = np.load(GOLD / "merge.npz")
    acc = np.zeros_like(g["merged"]); cnt = np.zeros(g["count"].shape)
    for (y0, x0), p in zip(g["tiles"], g["probs"]):
        acc[:, y0:y0 + 8, x0:x0 + 8] += p
        cnt[y0:y0 + 8, x0:x0 + 8] += 1
    m = np.where(cnt > 0, acc / np.maximum(cnt, 1), acc)
    assert np.allclose(m, g["merged"]) and np.array_equal(m.argmax(0).astype(np.uint8), g["argmax"])


def test_focal_loss_restatement_is_fastai_focal_loss():
    """oracle FocalLossFlat = fastai 2.5.1 FocalLoss.forward behind BaseLoss's transpose + flatten: (1 - exp(-ce))^gamma * ce, ce = weighted
    per-pixel cross-entropy, plain mean; gamma = 0 without weights is the cross-entropy"""
    import torch
    import torch.nn.functional as F
    from oracle import unet_oracle as O
    g = torch.Generator().manual_seed(0)
    z = torch.randn(2, 4, 5, 6, generator=g)
    y = torch.randint(0, 4, (2, 5, 6), generator=g)
    w = torch.tensor([0.4, 1.0, 2.0, 0.7])
    for gamma in (2.0, 0.5):
        ce = F.cross_entropy(z, y, weight=w, reduction="none")
        want = ((1 - torch.exp(-ce)) ** gamma * ce).mean()
        assert torch.allclose(O.FocalLossFlat(gamma=gamma, weight=w)(z, y), want, rtol=1e-6)
    assert torch.allclose(O.FocalLossFlat(gamma=0.0)(z, y), O.CrossEntropyLossFlat()(z, y), rtol=1e-6)
    f = O.FocalLossFlat()
    assert f.func.gamma == 2.0 and torch.equal(f.decodes(z), z.argmax(1)) and torch.allclose(f.activation(z), torch.softmax(z, 1))
