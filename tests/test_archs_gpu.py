"""The two remaining encoder constructors the reference imports (params_and_main.py:12): xresnet101 (bottleneck blocks, 23 in
stage 3) and xresnet34_deep (six stages: two more halvings, SIX UnetBlocks, skip indices [8, 7, 6, 5, 4, 2]) -- eval parity at the
north_star bar and strict per-tensor gradient parity on the smooth network (no ReLU sign can flip), against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)
from tests.test_configs_gpu import _normalise_head  # noqa: E402
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

