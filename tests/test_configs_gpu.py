"""Every BASELINE.json configuration at ITS OWN size against the CPU oracle, with north_star's literal bar
("per-pixel class logits within 1e-3 fp32, argmax masks bit-exact").

Truth for everything that two correct fp32 evaluations can disagree on (ReLU / max-pool sign flips of ~0 pre-activations in a
gradient, numerical ties of the two best logits in a mask) is an fp64 run of the SAME oracle: the HIP path must be as close
to fp64 as the fp32 CPU oracle is (or within the stated absolute bar), tensor by tensor / pixel by pixel.

  cfg2 (configs[1]): xresnet34 4->5, 512x512 -- full training step, EVERY parameter gradient per tensor (the narrow weight-gradient
        kernels wgrad_flat<7,5>/<6,7> and every conv launch at their real shapes); default-init tile with logits O(1):
        |dlogit| < 1e-3 absolute and identical masks;
  cfg1 (configs[0]): xresnet18 3->2, 256x256, batch 2 training step;
  cfg4 (configs[3]): xresnet50 8->10, ONE 1024x1024 tile, eval logits + mask; training step properties at that size;
  shipped default (reference params_and_main.py:36,83): 3-band 400x400 tiles, 3 classes, self-attention ON;
  SelfAttention at cfg2 size (4096 positions).
Reference call sites: train.py:247-250 (fit_one_cycle step), predict.py:193-203,232 (probabilities -> argmax).
"""
from unet_amd import ops as _knobs  # noqa: E402  (unet_tuning switches of this thread's launches)
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)


def _rel_l2(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()


def _hip_from(ref, arch, n_in, n_out, size, sa=False):
    from unet_amd.model import HipDynamicUnet
    model = HipDynamicUnet(arch, n_in, n_out, size, self_attention=sa)
    r = model.load_state_dict(ref.state_dict())
    assert not r.missing_keys and not r.unexpected_keys
    return model


def _normalise_head(ref, x, target=4.0):
    """scale the 1x1 head so that the eval logits are O(1) (max |z| = target): north_star's 1e-3 is an absolute bar on O(1) logits"""
    ref.eval()
    with torch.no_grad():
        s = ref(x).abs().max().item() / target
        head = ref.layers[-1][0]
        head.weight.div_(s)
        head.bias.div_(s)


def _assert_masks(am_hip, z_ref32, z_ref64, err_hip, err_cpu, what, max_ties=0):
    """north_star: argmax masks BIT-EXACT -- 0 differing pixels on every committed fixture (what has been measured on every one of them
    since round 2).  Should a pixel ever differ, the fp64 run of the oracle says what it is before the test fails: a numerical tie of the two
    best logits (fp64 margin below the fp32 evaluation error: undecidable in fp32) or a wrong decision"""
    am32, am64 = z_ref32.argmax(1), z_ref64.argmax(1)
    diff = am_hip != am32
    n = int(diff.sum())
    if n == 0:
        return 0
    top2 = z_ref64.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])
    print(f"{what}: {n} mask pixel(s) differ from the fp32 oracle; fp64 margins there {gap[diff].min():.2e} .. {gap[diff].max():.2e}, "
          f"fp32 evaluation errors hip {err_hip:.2e} cpu {err_cpu:.2e}; wrong vs fp64: hip {int((am_hip != am64).sum())} cpu {int((am32 != am64).sum())}")
    assert bool((gap[diff] <= 2.0 * max(err_hip, err_cpu)).all()), f"{what}: a mask pixel differs where fp64 is decided (gap {gap[diff].max():.2e})"
    wrong_hip, wrong_cpu = int((am_hip != am64).sum()), int((am32 != am64).sum())
    assert wrong_hip <= wrong_cpu + n, (what, wrong_hip, wrong_cpu)
    assert n <= max_ties, f"{what}: {n} tie pixels differ"
    return n


def _grad_table(model, ref, ref64):
    rows = []
    for (n, p), (n2, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        assert n == n2
        rows.append((n, _rel_l2(p.grad.cpu(), r.grad), _rel_l2(q.grad, r.grad), float(r.grad.abs().max())))
    return rows


# Why the encoder bound is looser than the decoder-tail bar.  The gradient of a ReLU / max-pool network is piecewise linear in the
# forward values: a pre-activation within rounding distance of 0 takes a different sign in two CORRECT fp32 evaluations and moves
# the gradients upstream of it by a finite amount.  The number of such flips is proportional to the forward rounding error, and
# that error differs between the two fp32 implementations for a known reason: a gfx950 fp32 MFMA accumulates its K = 9 Cin products
# as ONE k-ordered fmaf chain (cdna_hip_programming.md section 3), whose relative error grows like sqrt(K) -- measured against fp64
# on MI355X: 3.8e-7 at K = 891, 8.4e-7 at K = 4608, 1.1e-6 at K = 9216 for the forward conv, 1.6e-6 for the K = 9216 input gradient --
# while oneDNN sums in 16-lane blocks and stays at 1.7-2.5e-7 for every K (scripts/accuracy_vs_fp64.py).
# Both are far inside north_star's 1e-3 logit bar; the gradient inherits the ratio: measured e_hip / e_cpu32 is ~2 in the decoder
# and 4-9 in the deep encoder (K up to 9216 and few pixels per channel).  Hence: decoder tail absolute bar; everywhere else the HIP
# gradient may be at most ENC_FACTOR x as far from fp64 as the fp32 CPU oracle is on the same tensor, and never beyond ENC_CAP.
# Round 3: the deep stages at small batch now run as split-K launches (chains of K / splits products, partial sums added in fixed order):
# the forward error of the K = 4608 / 9216 layers dropped from 6.3e-7 / 9.4e-7 to 2.2e-7 / 2.4e-7 against fp64 (oneDNN: 2.5e-7), and the measured
# gradient ratios e_hip / e_cpu32 with it -- cfg2 max 3.2 (was 3.7), cfg1 3.9 (was 8.6), shipped default 3.3 -- so the factor is 5, not 12.
# (the absolute cap stays 5e-2: the six-stage xresnet34_deep reaches 3.6e-2 at a ratio of 1.7 -- its fp32 oracle is at 2.1e-2 itself)
ENC_FACTOR, ENC_CAP = 5.0, 5e-2


def _grad_stats(rows, tail_from, tail_bar, what, ec_ref=None):
    """rows: (name, e_hip, e_cpu32, scale) = per-tensor relative L2 distance to the fp64 oracle gradient.  Returns the three statistics
    the bars are set on: worst decoder-tail tensor (top-level child >= tail_from: hundreds of thousands of pixels average out single
    ReLU flips), worst EXCESS over max(tail_bar, ENC_FACTOR x e_cpu32) as a ratio (<= 1 = inside), worst absolute distance.
    ec_ref (several draws): the fp32 CPU oracle's distance on each tensor as the MEDIAN over the draws -- the flip-noise scale of the tensor,
    not the luck of the CPU run on this one draw (cfg1, measured: the CPU oracle's median tensor is 9.8e-4 from fp64 in one draw, 3.6e-3 and
    5.8e-3 in the other two; against the per-draw scale the first draw alone read 1.59 x the bar on one encoder BatchNorm weight)."""
    worst_tail, worst_rel, worst_abs = ("", 0.0), ("", 0.0, 0.0, 0.0), ("", 0.0)
    for i, (n, eh, ec, sc) in enumerate(rows):
        if sc == 0.0:
            continue
        if ec_ref is not None:
            ec = ec_ref[i]
        top = int(n.split(".")[1])
        if top >= tail_from and eh > worst_tail[1]:
            worst_tail = (n, eh)
        if eh > worst_abs[1]:
            worst_abs = (n, eh)
        ex = eh / max(tail_bar, ENC_FACTOR * ec)
        if ex > worst_rel[3]:
            worst_rel = (n, eh, ec, ex)
    print(f"{what}: worst tail {worst_tail}; median e_hip {sorted(r[1] for r in rows)[len(rows) // 2]:.2e} "
          f"median e_cpu32 {sorted(r[2] for r in rows)[len(rows) // 2]:.2e}; max e_hip {max(r[1] for r in rows):.2e}")
    ratios = sorted(r[1] / max(r[2], 1e-30) for r in rows if r[3] > 0)
    print(f"{what}: e_hip / e_cpu32 median {ratios[len(ratios) // 2]:.1f} max {ratios[-1]:.1f}; worst tensor {worst_abs}; worst excess {worst_rel}")
    return worst_tail, worst_rel, worst_abs


def _check_grads(rows, tail_from, tail_bar, what):
    """one draw: e_hip <= tail_bar on the decoder tail; every tensor e_hip <= max(tail_bar, ENC_FACTOR x e_cpu32) and <= ENC_CAP"""
    _check_grads_draws([rows], tail_from, tail_bar, what)


def _check_grads_draws(draws, tail_from, tail_bar, what):
    """Flip noise is a property of the DRAW (which pre-activations happen to sit within rounding distance of zero), not of the kernels: a
    kernel that gets CLOSER to fp64 moves which ones flip and can land on the wrong side of a single-draw bar (round 3: a +0.9 % change
    was taken back for 2.001e-3 against 2e-3).  With several draws (different tiles through the same network) the bars are asserted on
    the MEDIAN of three fixed draws -- always all three -- and no single draw may be beyond 1.25 x the bar (a real regression moves every
    draw; measured on MI355X, cfg2: worst excess 0.65 / 0.46 / 0.24 of the bar over the three draws).  The per-tensor noise scale in the
    relative bar is the CPU oracle's own distance to fp64 taken as the median over the draws (see _grad_stats)."""
    med = lambda v: sorted(v)[len(v) // 2]
    ec_ref = None if len(draws) == 1 else [med([rows[i][2] for rows in draws]) for i in range(len(draws[0]))]
    st = [_grad_stats(rows, tail_from, tail_bar, f"{what} draw {i}", ec_ref) for i, rows in enumerate(draws)]
    tails, exs, caps = [s_[0][1] for s_ in st], [s_[1][3] for s_ in st], [s_[2][1] for s_ in st]
    lim = 1.0 if len(draws) == 1 else 1.25
    assert med(tails) <= tail_bar and max(tails) <= lim * tail_bar, (what, "decoder tail", [s_[0] for s_ in st])
    assert med(exs) <= 1.0 and max(exs) <= lim, (what, "excess over max(tail_bar, ENC_FACTOR x e_cpu32)", [s_[1] for s_ in st])
    assert med(caps) <= ENC_CAP and max(caps) <= lim * ENC_CAP, (what, "absolute cap", [s_[2] for s_ in st])


# ------------------------------------------------------------------------------------------------ cfg2

def test_cfg2_training_step_every_gradient_against_the_oracle():
    """2 x (4 x 512 x 512), xresnet34, 5 classes, weighted CE, train mode (batch statistics), BatchNorm parameters randomised
    so that no path is trivially zero.  Every parameter gradient per tensor against fp64."""
    torch.manual_seed(0)
    ref = O.DynamicUnet("xresnet34", 4, 5, (512, 512))
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512))
    ref64 = copy.deepcopy(ref).double()
    w = torch.tensor([0.1, 0.3, 0.2, 0.25, 0.15])
    ref.train(); ref64.train(); model.train()
    draws = []
    for seed in (1234, 2024, 77):           # three draws of tiles through the same network, always all of them: the flip-noise bars are set on the median draw
        x, y = O.synthetic_batch(2, 4, 512, 512, 5, seed=seed)
        for m_ in (ref, ref64):
            m_.zero_grad()
        z32 = ref(x)
        l32 = O.CrossEntropyLossFlat(weight=w)(z32, y)
        l32.backward()
        z64 = ref64(x.double())
        l64 = O.CrossEntropyLossFlat(weight=w.double())(z64, y)
        l64.backward()
        loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
        torch.cuda.synchronize()
        z = model.logits_ts().view().permute(0, 3, 1, 2).cpu()
        scale = z64.abs().max().item()
        eh, ec = (z.double() - z64.detach()).abs().max().item(), (z32.detach().double() - z64.detach()).abs().max().item()
        print(f"train-mode logits (draw {seed}): scale {scale:.2f} err hip {eh:.2e} cpu32 {ec:.2e}; loss hip {loss.item():.7f} f64 {l64.item():.7f}")
        assert eh <= max(2e-6 * scale, 3.0 * ec)
        assert abs(loss.item() - l64.item()) <= 2e-6 * abs(l64.item())
        rows = _grad_table(model, ref, ref64)
        assert len(rows) == len(list(ref.parameters())) > 150
        draws.append(rows)
    _check_grads_draws(draws, tail_from=7, tail_bar=1e-3, what="cfg2 B=2")
    rows = draws[0]
    # the layers served by the narrow weight-gradient kernels, by name: final ResBlock 100->100 pair, last UnetBlock 192->96 / 96->96
    named = {n: (eh, ec) for n, eh, ec, _ in rows}
    for n in ("layers.11.convpath.0.0.weight", "layers.11.convpath.1.0.weight", "layers.7.conv1.0.weight", "layers.7.conv2.0.weight"):
        assert named[n][0] <= 1e-3, (n, named[n])
    # BatchNorm running statistics moved by the batch statistics
    for (n, b), (_, b2) in zip(model.named_buffers(), ref64.named_buffers()):
        if b.dtype.is_floating_point:
            assert (b.cpu().double() - b2).abs().max().item() <= 1e-5 * (1.0 + b2.abs().max().item()), n


def test_cfg2_smooth_network_every_gradient_tight():
    """The same geometry with every pre-activation pushed far from zero (no ReLU can flip, the network is a smooth function):
    now the comparison measures the backward KERNELS at their real shapes (16x16x4 conv / dgrad launches, wgrad_flat<7,5> and <6,7>,
    general / 1x1 / head weight-gradient kernels, BatchNorm backward, pooling and shuffle adjoints) and nothing else:
    every parameter gradient within 1e-4 relative L2 of the fp64 oracle, per tensor."""
    import torch.nn as nn
    torch.manual_seed(3)
    ref = O.DynamicUnet("xresnet34", 4, 5, (512, 512))
    O.randomize_bn_and_zero_gammas(ref, seed=4)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.bias.fill_(8.0)
            elif isinstance(m, nn.Conv2d) and m.bias is not None:
                m.weight.mul_(0.01)
                m.bias.fill_(1.0)
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512))
    ref64 = copy.deepcopy(ref).double()
    x, y = O.synthetic_batch(1, 4, 512, 512, 5)
    w = torch.tensor([0.1, 0.3, 0.2, 0.25, 0.15])
    ref.train(); ref64.train(); model.train()
    O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
    l64 = O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double()), y)
    l64.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l64.item()) <= 2e-6 * abs(l64.item())
    rows = _grad_table(model, ref, ref64)
    # a few BatchNorm biases / gammas of this fixture have gradients that are sums cancelling to ~0 (1e-47 .. 1e-20 in fp64, below
    # or at the fp32 rounding floor of their terms): no fp32 evaluation can hit them, the fp32 CPU oracle is off by 100 % too.  The
    # bar applies wherever the oracle itself is well conditioned in fp32 (its own distance to fp64 <= 1e-3)
    live = [r for r in rows if r[3] > 1e-30 and r[2] <= 1e-3]
    assert len(live) > 0.8 * len(rows)
    live.sort(key=lambda r: -r[1])
    print(f"cfg2 smooth: {len(live)} of {len(rows)} tensors; worst e_hip {[(r[0], f'{r[1]:.2e}', f'{r[2]:.2e}') for r in live[:4]]}")
    assert all(r[1] <= max(1e-4, 3.0 * r[2]) for r in live), [r for r in live if r[1] > max(1e-4, 3.0 * r[2])][:3]


def test_cfg2_default_init_tile_literal_bar():
    """The state bench.py trains from: fastai's default initialisation (BatchZero gammas, ICNR shuffle convs, kaiming decoder).
    One train-mode forward on a batch of 2 (moves the running statistics exactly as the first training step does), then the eval
    forward of one 512 x 512 tile: |dlogit| < 1e-3 ABSOLUTE on O(1) logits, probabilities within 1e-3, identical masks."""
    torch.manual_seed(11)
    ref = O.DynamicUnet("xresnet34", 4, 5, (512, 512))
    x, _ = O.synthetic_batch(2, 4, 512, 512, 5, seed=77)
    _normalise_head(ref, x[:1])
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512))
    ref64 = copy.deepcopy(ref).double()
    ref.train(); ref64.train(); model.train()
    with torch.no_grad():
        zt32, zt64 = ref(x), ref64(x.double())
        zt = model(x.cuda()).cpu()
    assert (zt - zt32).abs().max().item() < 1e-3 and (zt.double() - zt64).abs().max().item() < 1e-3
    ref.eval(); ref64.eval(); model.eval()
    with torch.no_grad():
        z32, z64 = ref(x[:1]), ref64(x[:1].double())
        probs, amax = model.predict_probs(x[:1].cuda())
        z = model(x[:1].cuda()).cpu()
    scale = z64.abs().max().item()
    err_hip, err_cpu = (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item()
    print(f"default init eval: logit scale {scale:.3f}, |hip-cpu32| {(z - z32).abs().max().item():.2e}, |hip-f64| {err_hip:.2e}, |cpu32-f64| {err_cpu:.2e}")
    assert 0.5 < scale < 50.0
    assert (z - z32).abs().max().item() < 1e-3 and err_hip < 1e-3
    assert (probs.cpu() - torch.softmax(z32, 1)).abs().max().item() < 1e-3
    n = _assert_masks(amax.cpu(), z32, z64, err_hip, err_cpu, "default init")
    print("default init: differing (fp64-adjudicated tie) pixels:", n)


def test_cfg2_randomised_bn_tile_literal_bar():
    """the BatchNorm-randomised fixture of test_fullsize_gpu with the head normalised to O(1) logits: absolute 1e-3, masks
    adjudicated by fp64"""
    torch.manual_seed(0)
    ref = O.DynamicUnet("xresnet34", 4, 5, (512, 512))
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    x, _ = O.synthetic_batch(4, 4, 512, 512, 5)
    x = x[3:4]
    _normalise_head(ref, x)
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512))
    ref64 = copy.deepcopy(ref).double()
    ref.eval(); ref64.eval(); model.eval()
    with torch.no_grad():
        z32, z64 = ref(x), ref64(x.double())
        probs, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    err_hip, err_cpu = (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item()
    print(f"randomised BN eval: |hip-cpu32| {(z - z32).abs().max().item():.2e}, |hip-f64| {err_hip:.2e}, |cpu32-f64| {err_cpu:.2e}")
    assert (z - z32).abs().max().item() < 1e-3 and err_hip < 1e-3
    assert (probs.cpu() - torch.softmax(z32, 1)).abs().max().item() < 1e-3
    _assert_masks(amax.cpu(), z32, z64, err_hip, err_cpu, "randomised BN")


# ------------------------------------------------------------------------------------------------ cfg1

def test_cfg1_xresnet18_rgb_256_batch2_training_step():
    """configs[0]: 3-channel 256 x 256 tiles, xresnet18, 2 classes, batch 2 (the reference's own CPU-runnable case)"""
    torch.manual_seed(5)
    ref = O.DynamicUnet("xresnet18", 3, 2, (256, 256))
    O.randomize_bn_and_zero_gammas(ref, seed=6)
    model = _hip_from(ref, "xresnet18", 3, 2, (256, 256))
    ref64 = copy.deepcopy(ref).double()
    w = torch.tensor([0.5, 0.5])
    ref.train(); ref64.train(); model.train()
    draws = []
    for seed in (1234, 77, 4242):           # three draws of tiles through the same network, always all of them: the flip-noise bars are set on the median draw
        x, y = O.synthetic_batch(2, 3, 256, 256, 2, seed=seed)
        for m_ in (ref, ref64):
            m_.zero_grad()
        O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
        z64 = ref64(x.double())
        l64 = O.CrossEntropyLossFlat(weight=w.double())(z64, y)
        l64.backward()
        loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
        torch.cuda.synchronize()
        z = model.logits_ts().view().permute(0, 3, 1, 2).cpu()
        assert (z.double() - z64.detach()).abs().max().item() < 1e-3 * max(1.0, z64.abs().max().item() / 8)
        assert abs(loss.item() - l64.item()) <= 5e-6 * abs(l64.item())
        draws.append(_grad_table(model, ref, ref64))
    _check_grads_draws(draws, tail_from=7, tail_bar=2e-3, what="cfg1 B=2")
    x, y = O.synthetic_batch(2, 3, 256, 256, 2)
    ref.eval(); ref64.eval(); model.eval()
    with torch.no_grad():
        z32, ze64 = ref(x), ref64(x.double())
        _, amax = model.predict_probs(x.cuda())
        ze = model(x.cuda()).cpu()
    assert (ze - z32).abs().max().item() < 1e-3 * max(1.0, z32.abs().max().item() / 8)
    n = _assert_masks(amax.cpu(), z32, ze64, (ze.double() - ze64).abs().max().item(), (z32.double() - ze64).abs().max().item(), "cfg1")
    print("cfg1: differing (fp64-adjudicated tie) pixels:", n)


# ------------------------------------------------------------------------------------------------ cfg4

@pytest.fixture(scope="module")
def cfg4():
    torch.manual_seed(2)
    ref = O.DynamicUnet("xresnet50", 8, 10, (1024, 1024))
    O.randomize_bn_and_zero_gammas(ref, seed=3)
    x, y = O.synthetic_batch(1, 8, 1024, 1024, 10)
    _normalise_head(ref, x[:, :, :256, :256])
    model = _hip_from(ref, "xresnet50", 8, 10, (1024, 1024))
    return ref, model, x, y


def test_cfg4_one_1024_tile_eval_against_the_oracle(cfg4):
    """configs[3]: 8-band 1024 x 1024 tile, xresnet50, 10 classes, fp32: 2048-channel bottleneck at 32 x 32, 392-wide final
    ResBlock at 1024 x 1024 (3.3 TFLOP), 10.15 TFLOP forward"""
    ref, model, x, _ = cfg4
    ref.eval(); model.eval()
    with torch.no_grad():
        z32 = ref(x)
        probs, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    err, scale = (z - z32).abs().max().item(), z32.abs().max().item()
    print(f"cfg4 eval: logit scale {scale:.2f}, |hip-cpu32| {err:.2e}")
    assert err < 1e-3 * max(1.0, scale / 8.0)
    assert (probs.cpu() - torch.softmax(z32, 1)).abs().max().item() < 1e-3
    # masks: an fp64 run of the same oracle decides every pixel the two fp32 evaluations disagree on (a million pixels, 10 classes)
    ref64 = copy.deepcopy(ref).double()
    ref64.eval()
    with torch.no_grad():
        z64 = ref64(x.double())
    del ref64
    err_hip, err_cpu = (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item()
    n = _assert_masks(amax.cpu(), z32, z64, err_hip, err_cpu, "cfg4")
    print(f"cfg4 eval: |hip-f64| {err_hip:.2e} |cpu32-f64| {err_cpu:.2e}; differing (fp64-adjudicated tie) pixels: {n}")


def test_cfg4_training_step_properties_at_full_size(cfg4):
    """one training step of the 1024 x 1024 xresnet50 tile: reproducible bit for bit, the two conv kernel families and the two
    3x3 weight-gradient families agree on the flat gradient, and the loss equals the oracle's train-mode loss"""
    from unet_amd._lib import lib
    ref, model, x, y = cfg4
    w = torch.full((10,), 0.1)
    xc, yc, wc = x.cuda(), y.cuda(), w.cuda()

    def step():
        model.train()
        model.flat_grad.zero_()
        l = model.forward_loss_backward(xc, yc, wc)
        torch.cuda.synchronize()
        return float(l.item()), model.flat_grad.clone()

    l0, g0 = step()
    l0b, g0b = step()
    assert l0 == l0b and torch.equal(g0, g0b)
    try:
        _knobs.set_knob("mfma_shape", 32)
        l1, g1 = step()
    finally:
        _knobs.set_knob("mfma_shape", 16)
    n0 = g0.double().norm().item()
    assert torch.isfinite(g0).all() and n0 > 0
    # a different conv kernel changes the summation order of every activation: ReLU sign flips of ~0 pre-activations move single
    # encoder gradients (batch of ONE tile: 1024 samples per channel at the bottleneck); the decoder span of the flat gradient
    # (everything from the post-encoder BatchNorm on) averages over >= 4096 pixels per channel and stays at rounding level
    do = model._decoder_offset
    e_all = (g1 - g0).double().norm().item() / n0
    e_dec = (g1[do:] - g0[do:]).double().norm().item() / g0[do:].double().norm().item()
    print(f"cfg4 train step: 16x16x4 vs 32x32x2 kernels: whole gradient {e_all:.2e}, decoder span {e_dec:.2e}")
    assert abs(l1 - l0) <= 1e-5 * abs(l0) and e_dec < 2e-3 and e_all < 3e-2
    ref.train()
    with torch.no_grad():
        l_ref = O.CrossEntropyLossFlat(weight=w)(ref(x), y).item()
    assert abs(l0 - l_ref) <= 2e-5 * abs(l_ref), (l0, l_ref)


# ------------------------------------------------------------------------------------------------ shipped default / SA

def _sa_pair(arch, n_in, n_out, size, seed, x, blk_idx=5, condition=True):
    """fastai's default initialisation (activations stay O(1): with randomised BatchNorm statistics the norm-free decoder inflates
    them to 1e4 and the attention logits f^T g to 1e8, where ONE fp32 ulp is 8 and softmax is a coin toss for either
    implementation), BatchNorm gammas of the ResBlock tails opened, running statistics moved by one train-mode pass, gamma = 0.7
    (it is 0 at init: the attention branch would not reach the logits)."""
    import torch.nn as nn
    torch.manual_seed(seed)
    ref = O.DynamicUnet(arch, n_in, n_out, size, self_attention=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, O.SelfAttention):
                m.gamma.fill_(0.7)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
        ref.train()
        ref(x)
        # bring the attention input to O(1): the decoder has no normalisation, and spectral norm fixes the projections' gain, so
        # the attention logits f^T g scale with the SQUARE of the activation magnitude (1e5 here: one fp32 ulp = 0.008 in the
        # exponent of the softmax).  Scale the conv that feeds the attention block instead.
        seen = {}
        blk = ref.layers[blk_idx]
        h = blk.conv2[1].register_forward_hook(lambda m, i, o: seen.__setitem__("a", float(o.abs().max())))
        ref(x)
        h.remove()
        if condition:
            blk.conv2[0].weight.div_(seen["a"])
            blk.conv2[0].bias.div_(seen["a"])
            ref(x)                                  # running statistics downstream follow the new scale
    return ref


def _attention_logit_scale(ref, x, blk_idx=5):
    """max |f^T g| the oracle's attention block sees for x (eval mode): the conditioning of its softmax"""
    sa = ref.layers[blk_idx].conv2[2]
    seen = {}
    h = sa.register_forward_hook(lambda m, i, o: seen.__setitem__("x", i[0]))
    with torch.no_grad():
        ref(x)
        h.remove()
        xi = seen["x"].flatten(2)
        return float(torch.bmm(sa.query(xi).transpose(1, 2), sa.key(xi)).abs().max())


def test_shipped_default_400px_rgb_3class_self_attention_on():
    """reference params_and_main.py:36,49,83,101: patch_size 400, 3 bands, 3 classes, xresnet34, self_attention = True, batch 4
    (here 2): SelfAttention(384) on the 50 x 50 stage (2500 positions), nearest-resize paths (400 is not divisible by 32)"""
    x, y = O.synthetic_batch(2, 3, 400, 400, 3)
    ref = _sa_pair("xresnet34", 3, 3, (400, 400), 21, x)
    _normalise_head(ref, x[:1])
    model = _hip_from(ref, "xresnet34", 3, 3, (400, 400), sa=True)
    ref.eval(); model.eval()
    with torch.no_grad():
        z32 = ref(x)
        _, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    err = (z - z32).abs().max().item()
    print(f"shipped default eval: |hip-cpu32| {err:.2e} at scale {z32.abs().max().item():.2f}; attention logits up to {_attention_logit_scale(ref, x):.1f}")
    assert err < 1e-3
    ref64 = copy.deepcopy(ref).double()
    ref64.eval()
    with torch.no_grad():
        z64 = ref64(x.double())
    _assert_masks(amax.cpu(), z32, z64, (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item(), "shipped default")
    # training step: loss + every gradient against fp64 (incl. gamma and the spectral-normed projections)
    ref.train(); ref64.train(); model.train()
    w = torch.tensor([0.2, 0.5, 0.3])
    O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
    l64 = O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double()), y)
    l64.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l64.item()) <= 5e-6 * abs(l64.item())
    rows = _grad_table(model, ref, ref64)
    _check_grads(rows, tail_from=7, tail_bar=2e-3, what="shipped default")
    sa_rows = [r for r in rows if ".conv2.2." in r[0]]
    assert len(sa_rows) == 4 and all(r[1] <= max(2e-3, ENC_FACTOR * r[2]) for r in sa_rows), sa_rows


def test_self_attention_at_cfg2_size_4096_positions():
    """cfg2 geometry with self_attention = True: SelfAttention(384) on the 64 x 64 stage = 4096 positions (64 MiB of attention
    weights per tile in the oracle)"""
    x, y = O.synthetic_batch(1, 4, 512, 512, 5)
    ref = _sa_pair("xresnet34", 4, 5, (512, 512), 31, x)
    _normalise_head(ref, x)
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512), sa=True)
    ref.eval(); model.eval()
    with torch.no_grad():
        z32 = ref(x)
        _, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    err = (z - z32).abs().max().item()
    print(f"SA 4096 eval: |hip-cpu32| {err:.2e}; attention logits up to {_attention_logit_scale(ref, x):.1f}")
    assert err < 1e-3
    ref64 = copy.deepcopy(ref).double()
    ref64.eval()
    with torch.no_grad():
        z64 = ref64(x.double())
    _assert_masks(amax.cpu(), z32, z64, (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item(), "SA 4096")
    ref.train(); ref64.train(); model.train()
    O.CrossEntropyLossFlat()(ref(x), y).backward()
    O.CrossEntropyLossFlat()(ref64(x.double()), y).backward()
    model.forward_loss_backward(x.cuda(), y.cuda(), None)
    torch.cuda.synchronize()
    rows = [r for r in _grad_table(model, ref, ref64) if ".conv2.2." in r[0]]     # gamma, query / key / value weight_orig
    print("SA 4096 gradients (name, e_hip, e_cpu32):", [(r[0].split("conv2.2.")[1], f"{r[1]:.2e}", f"{r[2]:.2e}") for r in rows])
    assert len(rows) == 4 and all(r[1] <= max(2e-3, ENC_FACTOR * r[2]) for r in rows), rows



def test_self_attention_at_the_unconditioned_scale_is_adjudicated_by_fp64():
    """The two SA fixtures above scale the conv in front of the attention block so that the attention logits f^T g stay O(100).  Without
    that step the norm-free decoder feeds activations of O(100) and the logits reach 1e4, where one fp32 ulp of a logit is 1e-3 in the
    exponent of the softmax: two correct fp32 evaluations then differ visibly, and the question is only whether the HIP path is as close to
    the truth as the fp32 CPU oracle is.  An fp64 run of the same oracle is the truth: the HIP logits must be no further from it than a
    small multiple of the fp32 oracle's own distance."""
    x, _ = O.synthetic_batch(1, 4, 512, 512, 5)
    ref = _sa_pair("xresnet34", 4, 5, (512, 512), 31, x, condition=False)
    _normalise_head(ref, x)
    model = _hip_from(ref, "xresnet34", 4, 5, (512, 512), sa=True)
    ref64 = copy.deepcopy(ref).double()
    ref.eval(); ref64.eval(); model.eval()
    with torch.no_grad():
        z32, z64 = ref(x), ref64(x.double())
        _, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    scale = _attention_logit_scale(ref, x)
    e_hip, e_cpu = (z.double() - z64).abs().max().item(), (z32.double() - z64).abs().max().item()
    r_hip, r_cpu = _rel_l2(z, z64), _rel_l2(z32, z64)
    print(f"SA unconditioned: attention logits up to {scale:.0f}; max |hip-f64| {e_hip:.2e} |cpu32-f64| {e_cpu:.2e}; rel L2 hip {r_hip:.2e} cpu32 {r_cpu:.2e}")
    assert scale > 1e3                                      # the fixture really is ill conditioned
    # Measured on MI355X: rel L2 6.0e-6 (HIP) against 2.2e-6 (fp32 CPU oracle), max 6.8e-4 against 1.4e-4 -- both five orders below the
    # logits.  Before the split-K launches the HIP path stood at 1.7e-5 / 1.8e-3: a gfx950 fp32 MFMA sums a reduction as ONE k-ordered chain
    # (rounding error ~ sqrt(K)) where oneDNN keeps 16 partial sums; the attention product O = P H reduces over 4096 positions and now runs
    # as 4 chains of 1024.  The HIP path is as close to fp64 as the fp32 oracle is, to within a small factor.
    assert r_hip <= 4.0 * r_cpu + 1e-7 and r_hip <= 1e-4 and e_hip <= 8.0 * e_cpu + 1e-6
    # masks against the fp64 decision: the HIP mask may be wrong only where fp32 cannot decide, and no more often than the fp32 oracle
    am64 = z64.argmax(1)
    top2 = z64.topk(2, dim=1).values
    gap = top2[:, 0] - top2[:, 1]
    wrong_hip, wrong_cpu = amax.cpu() != am64, z32.argmax(1) != am64
    assert bool((gap[wrong_hip] <= 2.0 * max(e_hip, e_cpu)).all())
    assert int(wrong_hip.sum()) <= 2 * int(wrong_cpu.sum()) + 4, (int(wrong_hip.sum()), int(wrong_cpu.sum()))


def test_cfg4_smooth_network_every_gradient_at_1024():
    """configs[3] geometry, ONE 1024 x 1024 xresnet50 tile, every pre-activation pushed far from zero (no ReLU can flip): the backward
    KERNELS at their cfg4 shapes -- 2048- / 4096-channel bottleneck weight gradients, wgrad1x1 at 1024 x 1024, the 392-wide final ResBlock --
    against the fp32 CPU oracle, per tensor.  Both sides are fp32, so the bar is a few fp32 roundings of a long reduction (up to 1024^2
    pixels per weight): 5e-4 relative L2; tensors whose gradient is a sum cancelling to ~0 are skipped (the oracle cannot hit them either)."""
    import torch.nn as nn
    torch.manual_seed(7)
    ref = O.DynamicUnet("xresnet50", 8, 10, (1024, 1024))
    O.randomize_bn_and_zero_gammas(ref, seed=8)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.bias.fill_(8.0)
            elif isinstance(m, nn.Conv2d) and m.bias is not None:
                m.weight.mul_(0.01)
                m.bias.fill_(1.0)
    model = _hip_from(ref, "xresnet50", 8, 10, (1024, 1024))
    x, y = O.synthetic_batch(1, 8, 1024, 1024, 10)
    w = torch.rand(10) + 0.5
    ref.train(); model.train()
    l32 = O.CrossEntropyLossFlat(weight=w)(ref(x), y)
    l32.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l32.item()) <= 1e-5 * abs(l32.item())
    norms = {n: float(q.grad.double().norm()) for n, q in ref.named_parameters()}
    dims = {n: p.dim() for n, p in ref.named_parameters()}
    rows, bad = [], []
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu().double() - q.grad.double()).norm())
        if norms[n] == 0.0:
            continue
        if dims[n] == 4:
            # conv filters (the weight-gradient kernels: sums of products over up to 2^20 pixels): 5e-4 relative L2
            den, bar = norms[n], 5e-4
        else:
            # per-channel vectors (conv biases, BatchNorm gamma / beta) are plain sums of a gradient tensor that largely cancel -- the head
            # bias sums softmax - onehot over a million pixels, the BatchZero betas of this fixture cancel to 1e-20 .. 1e-40 -- and both fp32
            # evaluations lose digits to the cancellation: 5e-3, measured against the vector's own norm or, where it has cancelled away,
            # against 1e-3 of the filter gradient of the same ConvLayer (conv, BatchNorm, ReLU)
            parent = n.rsplit(".", 2)[0] + ".0.weight"
            den, bar = max(norms[n], 1e-3 * norms.get(parent, 0.0) if parent != n else 0.0), 5e-3
        rows.append((n, err / den, norms[n]))
        if err > bar * den:
            bad.append(rows[-1])
    rows.sort(key=lambda r: -r[1])
    print(f"cfg4 smooth: {len(rows)} tensors ({sum(1 for n in dims if dims[n] == 4)} conv filters); worst {[(r[0], f'{r[1]:.2e}', f'{r[2]:.1e}') for r in rows[:5]]}")
    assert len(rows) > 0.9 * len(dims)
    assert not bad, bad[:5]
