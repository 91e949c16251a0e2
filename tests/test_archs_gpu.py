"""The two remaining encoder constructors the reference imports (params_and_main.py:12): xresnet101 (bottleneck blocks, 23 in
stage 3) and xresnet34_deep (six stages: two more halvings, SIX UnetBlocks, skip indices [8, 7, 6, 5, 4, 2]) -- eval parity at the
north_star bar and strict per-tensor gradient parity on the smooth network (no ReLU sign can flip), against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)
from tests.test_configs_gpu import (ENC_FACTOR, _attention_logit_scale, _check_grads, _grad_table, _hip_from, _normalise_head,  # noqa: E402
                                     _sa_pair)
from tests.test_model_gpu import _make_all_active  # noqa: E402

ARCHS = [
    ("xresnet101", 3, 4, (64, 64), 2),
    ("xresnet34_deep", 4, 3, (256, 256), 2),
    ("xresnet34_deep", 3, 2, (160, 224), 1),     # not divisible by 128: ceil-mode pooling and nearest resize in the two deepest blocks
]


@pytest.mark.parametrize("arch,n_in,n_out,size,bs", ARCHS)
def test_structure_and_eval(arch, n_in, n_out, size, bs):
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    x, _ = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    _normalise_head(ref, x)          # 33 randomised bottleneck blocks inflate the activations: the 1e-3 bar is for O(1) logits
    model = HipDynamicUnet(arch, n_in, n_out, size)
    missing = model.load_state_dict(ref.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(model.sz_chg_idxs) == list(ref.sz_chg_idxs)
    assert [n for n, _ in model.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert sum(p.numel() for p in model.parameters()) == sum(p.numel() for p in ref.parameters())
    # the fastai splitter (train.py:78-80) yields the same three groups
    g_ref = [sum(p.numel() for p in g) for g in O.xresnet_split(ref)]
    from unet_amd.optimizer import FlatAdam
    opt = FlatAdam(model, [1e-5, 3e-5, 1e-4])
    assert [sum(p.numel() for p in g) for g in opt.groups] == g_ref
    ref.eval(); model.eval()
    with torch.no_grad():
        z_ref = ref(x)
        z = model(x.cuda()).cpu()
    assert z.shape == z_ref.shape
    err = (z - z_ref).abs().max().item()
    assert err < 1e-3, f"logit err {err}"
    # masks: identical except where the oracle's own top-2 margin is below the logit error (a tie in fp32)
    m, m_ref = z.argmax(1), z_ref.argmax(1)
    top2 = z_ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    assert bool(((m == m_ref) | (margin <= 2 * err)).all())
    assert (m != m_ref).float().mean().item() < 1e-4


@pytest.mark.parametrize("arch,n_in,n_out,size,bs", ARCHS[:2])
def test_train_step_gradients_smooth(arch, n_in, n_out, size, bs):
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(3)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=4)
    _make_all_active(ref)
    model = HipDynamicUnet(arch, n_in, n_out, size)
    model.load_state_dict(ref.state_dict())
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    w = torch.rand(n_out) + 0.5
    ref.train(); model.train()
    loss_ref = O.CrossEntropyLossFlat(weight=w)(ref(x), y)
    loss_ref.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    worst = ("", 0.0)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        e = (p.grad.cpu() - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-12)
        if e > worst[1]:
            worst = (n, e)
    print("smooth worst", worst)
    assert worst[1] < 2e-3, worst


def test_deep_encoder_with_self_attention():
    """reference params_and_main.py:12,83: xresnet34_deep is an importable ARCHITECTURE and self_attention = True the shipped default.
    fastai puts SelfAttention on UnetBlock len - 3 = 3 of its six blocks: 432 channels (query / key 54 wide, padded to 56 lanes in the
    fused QKV buffer) on the 32 x 32 stage of a 256 x 256 tile.  Eval logits / masks and every gradient -- gamma and the spectral-normed
    projections included -- against the oracle (fp64 adjudicates)."""
    import copy
    x, y = O.synthetic_batch(2, 3, 256, 256, 3)
    ref = _sa_pair("xresnet34_deep", 3, 3, (256, 256), 41, x, blk_idx=7)
    assert ref.layers[7].conv2[2].query[0].weight.shape[:2] == (54, 432)
    _normalise_head(ref, x[:1])
    model = _hip_from(ref, "xresnet34_deep", 3, 3, (256, 256), sa=True)
    ref.eval(); model.eval()
    with torch.no_grad():
        z32 = ref(x)
        _, amax = model.predict_probs(x.cuda())
        z = model(x.cuda()).cpu()
    err = (z - z32).abs().max().item()
    print(f"deep + SA eval: |hip-cpu32| {err:.2e} at scale {z32.abs().max().item():.2f}; attention logits up to {_attention_logit_scale(ref, x, 7):.1f}")
    assert err < 1e-3
    diff = amax.cpu() != z32.argmax(1)
    top2 = z32.topk(2, dim=1).values
    assert int(diff.sum()) <= 2 and bool(((top2[:, 0] - top2[:, 1])[diff] <= 4 * err).all())
    ref64 = copy.deepcopy(ref).double()
    ref.train(); ref64.train(); model.train()
    w = torch.tensor([0.2, 0.5, 0.3])
    O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
    l64 = O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double()), y)
    l64.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - l64.item()) <= 5e-6 * abs(l64.item())
    rows = _grad_table(model, ref, ref64)
    _check_grads(rows, tail_from=9, tail_bar=2e-3, what="xresnet34_deep + SA")
    sa_rows = [r for r in rows if ".conv2.2." in r[0]]
    assert len(sa_rows) == 4 and all(r[1] <= max(2e-3, ENC_FACTOR * r[2]) for r in sa_rows), sa_rows
