import sys, torch, copy
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
import test_model_gpu as T
torch.set_num_threads(16)
for (arch, n_in, n_out, size, bs) in (("xresnet18", 3, 2, (96, 64), 2),):
    torch.manual_seed(3)
    ref = O.DynamicUnet(arch, n_in, n_out, size); O.randomize_bn_and_zero_gammas(ref, seed=4); T._make_bimodal(ref)
    ref64 = copy.deepcopy(ref).double()
    model = HipDynamicUnet(arch, n_in, n_out, size); model.load_state_dict(ref.state_dict())
    x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
    w = torch.rand(n_out) + 0.5
    ref.train(); model.train(); ref64.train()
    t32, t64 = {}, {}
    O.CrossEntropyLossFlat(weight=w)(ref(x, t32), y).backward()
    O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double(), t64), y).backward()
    model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda()); torch.cuda.synchronize()
    for k in t64:
        a = t64[k]; near = (a.abs() < 1e-3) & (a != 0)
        print("tap", k, "min |nonzero|", float(a[a != 0].abs().min()) if (a != 0).any() else None, "zeros frac", float((a == 0).float().mean()))
    for (n, p), (_, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        s = r.grad.abs().max().item()
        if s == 0: continue
        eh = (p.grad.cpu().double() - r.grad).abs().max().item() / s
        ec = (q.grad.double() - r.grad).abs().max().item() / s
        if eh > 1e-3 and ec < 1e-3:
            d = (p.grad.cpu().double() - r.grad).abs().flatten(1).max(1).values if r.grad.dim() > 1 else (p.grad.cpu().double() - r.grad).abs()
            rm = r.grad.abs().flatten(1).max(1).values if r.grad.dim() > 1 else r.grad.abs()
            top = d.topk(min(4, d.numel()))
            print(f"{n:36s} scale {s:.2e} e_hip {eh:.2e} e_cpu {ec:.2e}; worst out-channels {top.indices.tolist()} err {[f'{v:.1e}' for v in top.values.tolist()]} ref {[f'{rm[i]:.1e}' for i in top.indices.tolist()]} ; n zero-ref channels {(rm==0).sum().item()} of {rm.numel()}, hip nonzero there: {int(((rm==0) & (d>0)).sum())}")
