import sys, torch, copy
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet

def run(arch, n_in, n_out, size, bs):
    torch.manual_seed(0)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    model = HipDynamicUnet(arch, n_in, n_out, size)
    model.load_state_dict(ref.state_dict())
    ref64 = copy.deepcopy(ref).double()
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    ref.train(); model.train(); ref64.train()
    O.CrossEntropyLossFlat()(ref(x), y).backward()
    O.CrossEntropyLossFlat()(ref64(x.double()), y).backward()
    model.forward_loss_backward(x.cuda(), y.cuda(), None)
    torch.cuda.synchronize()
    rows = []
    for (n, p), (_, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        s = r.grad.abs().max().item() + 1e-30
        rows.append((n, (p.grad.cpu().double() - r.grad).abs().max().item() / s, (q.grad.double() - r.grad).abs().max().item() / s, s))
    print(arch, size, bs)
    for n, eh, ec, s in reversed(rows):
        if not n.startswith('layers.0.') or eh > 10 * ec:
            print('  %-44s hip %.2e cpu %.2e scale %.2e %s' % (n, eh, ec, s, '<<<' if eh > 20 * ec and eh > 1e-4 else ''))

run('xresnet18', 3, 2, (80, 80), 1)
run('xresnet34', 4, 5, (64, 64), 1)
