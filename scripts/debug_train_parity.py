import sys, torch, copy
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet

def run(arch, n_in, n_out, size, bs, weighted=True):
    torch.manual_seed(0)
    ref = O.DynamicUnet(arch, n_in, n_out, size)
    O.randomize_bn_and_zero_gammas(ref, seed=1)
    model = HipDynamicUnet(arch, n_in, n_out, size)
    model.load_state_dict(ref.state_dict())
    ref64 = copy.deepcopy(ref).double()
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    w = (torch.rand(n_out) + 0.5) if weighted else None
    ref.train(); model.train(); ref64.train()
    z_ref = ref(x)
    loss_ref = O.CrossEntropyLossFlat(weight=w)(z_ref, y)
    loss_ref.backward()
    z64 = ref64(x.double())
    O.CrossEntropyLossFlat(weight=None if w is None else w.double())(z64, y).backward()
    loss = model.forward_loss_backward(x.cuda(), y.cuda(), None if w is None else w.cuda())
    torch.cuda.synchronize()
    z = model.logits_ts().view().permute(0, 3, 1, 2).cpu()
    print(arch, size, bs, 'logit err hip-vs-cpu', (z - z_ref.detach()).abs().max().item(), 'hip-vs-f64', (z.double() - z64.detach()).abs().max().item(),
          'cpu-vs-f64', (z_ref.detach().double() - z64.detach()).abs().max().item(), 'loss', loss.item(), loss_ref.item())
    rows = []
    for (n, p), (_, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        s = r.grad.abs().max().item() + 1e-30
        e_hip = (p.grad.cpu().double() - r.grad).abs().max().item() / s
        e_cpu = (q.grad.double() - r.grad).abs().max().item() / s
        rows.append((e_hip, e_cpu, n, s))
    rows.sort(reverse=True)
    for e_hip, e_cpu, n, s in rows[:5]:
        print('  grad %-40s hip-vs-f64 %.3e  cpu32-vs-f64 %.3e  scale %.3e' % (n, e_hip, e_cpu, s))
    print('  median hip err %.3e  median cpu err %.3e' % (sorted(r[0] for r in rows)[len(rows)//2], sorted(r[1] for r in rows)[len(rows)//2]))

run('xresnet34', 4, 5, (64, 64), 1)
run('xresnet34', 4, 5, (64, 64), 2)
run('xresnet34', 4, 5, (128, 128), 2)
run('xresnet18', 3, 2, (80, 80), 1)
run('xresnet50', 8, 10, (64, 64), 1)
run('xresnet50', 8, 10, (128, 128), 2)
