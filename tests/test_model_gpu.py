"""Whole-network parity of the HIP path against the CPU oracle (same weights, same tiles).

Bar (BASELINE.json north_star): per-pixel class logits within 1e-3 in fp32, argmax masks identical.
Gradients: every parameter gradient within 2e-3 of that tensor's max magnitude (fp32 summation order
differs: MFMA k-order / split-K partials vs oneDNN)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O  # noqa: E402  (checker)


def _pair(arch, n_in, n_out, size, seed=0):
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(seed)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=seed + 1)
    model = HipDynamicUnet(arch, n_in, n_out, size)
    missing = model.load_state_dict(ref.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref, model


CASES = [
    ("xresnet34", 4, 5, (64, 64), 1),
    ("xresnet34", 4, 5, (64, 64), 2),
    ("xresnet18", 3, 2, (80, 80), 1),     # not divisible by 32: ceil-mode pooling + nearest resize paths
    ("xresnet18", 3, 2, (96, 64), 2),
    ("xresnet50", 8, 10, (64, 64), 1),
]


@pytest.mark.parametrize("arch,n_in,n_out,size,bs", CASES)
def test_eval_logits_and_masks(arch, n_in, n_out, size, bs):
    ref, model = _pair(arch, n_in, n_out, size)
    x, _ = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    ref.eval(); model.eval()
    with torch.no_grad():
        z_ref = ref(x)
        z = model(x.cuda())
    torch.cuda.synchronize()
    z = z.cpu()
    assert z.shape == z_ref.shape
    err = (z - z_ref).abs().max().item()
    assert err < 1e-3, f"logit err {err}"
    probs, amax = model.predict_probs(x.cuda())
    pr_ref = torch.softmax(z_ref, dim=1)
    assert (probs.cpu() - pr_ref).abs().max().item() < 1e-3
    assert torch.equal(amax.cpu(), pr_ref.argmax(dim=1)), "argmax masks differ"


@pytest.mark.parametrize("arch,n_in,n_out,size,bs", CASES)
def test_train_step_gradients(arch, n_in, n_out, size, bs):
    ref, model = _pair(arch, n_in, n_out, size)
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    w = torch.rand(n_out) + 0.5
    ref.train(); model.train()
    z_ref = ref(x)
    loss_ref = O.CrossEntropyLossFlat(weight=w)(z_ref, y)
    loss_ref.backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda())
    torch.cuda.synchronize()
    z = model.logits_ts().view().permute(0, 3, 1, 2).cpu()
    assert (z - z_ref.detach()).abs().max().item() < 1e-3
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    worst = ("", 0.0)
    for (n, p), (n2, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == n2
        scale = q.grad.abs().max().item()
        e = (p.grad.cpu() - q.grad).abs().max().item() / (scale + 1e-9)
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < 2e-3, f"worst gradient mismatch {worst}"
    # BatchNorm running statistics follow the batch statistics
    for (n, b), (_, b2) in zip(model.named_buffers(), ref.named_buffers()):
        if b.dtype.is_floating_point:
            assert (b.cpu() - b2).abs().max().item() < 1e-4 * (1 + b2.abs().max().item()), n
        else:
            assert int(b.item()) == int(b2.item()), n


def test_autograd_bridge_matches_fused_path():
    """loss.backward() through torch autograd (fastai-style loop) == the fused CE path."""
    ref, model = _pair("xresnet18", 4, 5, (64, 64))
    x, y = O.synthetic_batch(2, 4, 64, 64, 5)
    model.train()
    model.forward_loss_backward(x.cuda(), y.cuda(), None)
    g1 = model.flat_grad.clone()
    logits = model(x.cuda())
    assert logits.requires_grad and logits.shape == (2, 5, 64, 64)
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    model.flat_grad.zero_()
    loss.backward()
    torch.cuda.synchronize()
    err = (model.flat_grad - g1).abs().max().item() / g1.abs().max().item()
    assert err < 1e-4, err


def test_adam_step_and_second_forward():
    """one fastai-Adam step on the flat buffer == the oracle's per-tensor step; packed filters are refreshed."""
    from unet_amd.optimizer import FlatAdam
    ref, model = _pair("xresnet18", 4, 5, (64, 64))
    x, y = O.synthetic_batch(2, 4, 64, 64, 5)
    lrs = list(O.even_mults(1e-3 / 10, 1e-3, 3))
    opt_ref = O.FastaiAdam(O.xresnet_split(ref), lrs, no_wd=O.bn_bias_params(ref))
    opt = FlatAdam(model, lrs)
    ref.train(); model.train()
    for step in range(2):
        opt_ref.zero_grad()
        O.CrossEntropyLossFlat()(ref(x), y).backward()
        opt_ref.step()
        model.forward_loss_backward(x.cuda(), y.cuda(), None)
        opt.step()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        worst = max(worst, (p.detach().cpu() - q.detach()).abs().max().item())
    assert worst < 2e-4, worst    # |update| <= lr = 1e-3 per step; sign flips of ~0 gradients stay below this
    ref.eval(); model.eval()
    with torch.no_grad():
        err = (model(x.cuda()).cpu() - ref(x)).abs().max().item()
    assert err < 5e-3, err


def test_state_dict_roundtrip_and_indexing():
    ref, model = _pair("xresnet34", 4, 5, (64, 64))
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(sd[k].cpu(), v), k
    assert sum(p.numel() for p in model.parameters()) == 41244577
    # fastai splitter contract (train.py:78-80)
    n0 = sum(p.numel() for l in model[0][:3] for p in l.parameters())
    n1 = sum(p.numel() for l in model[0][3:] for p in l.parameters())
    n2 = sum(p.numel() for l in model[1:] for p in l.parameters())
    assert (n0, n1, n2) == (29056, 21275136, 19940385)
