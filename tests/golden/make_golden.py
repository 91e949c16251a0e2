"""Generates tests/golden/*.npz from the CPU oracle (oracle/unet_oracle.py).

Run from the repo root:  python tests/golden/make_golden.py
The reference repository holds no fixtures for this path and fastai cannot be imported here, so
these vectors pin (a) the oracle against drift and (b) the HIP path against the oracle on the GPU box, where
/root/reference does not exist.  Weights are re-created from the seed (41 M parameters do not belong in git);
the fixture stores inputs, outputs and a parameter checksum that guards the seeding."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import unet_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent


def build(arch, n_in, n_out, size, seed):
    torch.manual_seed(seed)
    m = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(m, seed=seed + 1)
    return m


def checksum(m):
    return float(sum(p.double().abs().sum() for p in m.parameters()))


def net_case(name, arch, n_in, n_out, size, bs, seed):
    m = build(arch, n_in, n_out, size, seed)
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out, seed=seed + 2)
    w = torch.rand(n_out, generator=torch.Generator().manual_seed(seed + 3)) + 0.5
    m.eval()
    with torch.no_grad():
        z_eval = m(x)
    m.train()
    z_train = m(x)
    loss = O.CrossEntropyLossFlat(weight=w)(z_train, y)
    loss.backward()
    L = m.layers
    grads = {"head_w": L[12][0].weight.grad, "head_b": L[12][0].bias.grad, "res1_w": L[11].convpath[0][0].weight.grad[:8],
             "shuf8_w": L[8][0][0].weight.grad[:16], "u3_conv2_b": L[7].conv2[0].bias.grad}
    np.savez_compressed(OUT / f"{name}.npz", arch=arch, n_in=n_in, n_out=n_out, size=np.array(size), bs=bs, seed=seed,
                        x=x.numpy(), y=y.numpy(), w=w.numpy(), z_eval=z_eval.numpy(), z_train=z_train.detach().numpy(),
                        loss=float(loss), param_checksum=checksum(m), argmax_eval=z_eval.argmax(1).numpy().astype(np.uint8),
                        **{f"g_{k}": v.numpy() for k, v in grads.items()})
    print(name, "loss", float(loss), "checksum", checksum(m))


def merge_case():
    """predict.py:284-326: overlapping tiles -> sum of probabilities + hit counter -> divide -> argmax."""
    g = torch.Generator().manual_seed(5)
    tiles = [(0, 0), (0, 5), (4, 2)]
    probs = [torch.softmax(torch.randn(3, 8, 8, generator=g), dim=0) for _ in tiles]
    MH, MW = 12, 13
    acc = torch.zeros(3, MH, MW); cnt = torch.zeros(MH, MW)
    for (y0, x0), p in zip(tiles, probs):
        acc[:, y0:y0 + 8, x0:x0 + 8] += p
        cnt[y0:y0 + 8, x0:x0 + 8] += 1
    merged = torch.where(cnt > 0, acc / cnt.clamp(min=1), acc)
    np.savez_compressed(OUT / "merge.npz", tiles=np.array(tiles), probs=torch.stack(probs).numpy(), merged=merged.numpy(),
                        argmax=merged.argmax(0).numpy().astype(np.uint8), count=cnt.numpy().astype(np.int32))


def optim_case():
    """two fastai-Adam steps + one-cycle schedule values on a 3-group toy problem"""
    g = torch.Generator().manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(7, generator=g)) for _ in range(3)]
    p0 = [p.detach().clone() for p in ps]
    lr_f, mom_f = O.one_cycle_scheds(O.even_mults(1e-3 / 10, 1e-3, 3))
    opt = O.FastaiAdam([[p] for p in ps], lr_f(0.0), no_wd=[ps[1]])
    grads = []
    for it, pct in enumerate((0.0, 0.5)):
        opt.lrs, opt.mom = list(lr_f(pct)), float(mom_f(pct))
        gr = [torch.randn(7, generator=g) for _ in ps]
        grads.append(torch.stack(gr))
        for p, gg in zip(ps, gr):
            p.grad = gg.clone()
        opt.step()
    pcts = np.array([0.0, 0.1, 0.25, 0.5, 0.9, 1.0])
    np.savez_compressed(OUT / "optim.npz", p0=torch.stack(p0).numpy(), grads=torch.stack(grads).numpy(),
                        p2=torch.stack([p.detach() for p in ps]).numpy(), pcts=pcts,
                        lr=np.stack([lr_f(float(t)) for t in pcts]), mom=np.array([mom_f(float(t)) for t in pcts]))


if __name__ == "__main__":
    net_case("net_x34_4to5_64", "xresnet34", 4, 5, (64, 64), 2, seed=100)
    net_case("net_x18_3to2_80", "xresnet18", 3, 2, (80, 80), 1, seed=200)
    merge_case()
    optim_case()
