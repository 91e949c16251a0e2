import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import unet_oracle as O
from unet_amd.model import HipDynamicUnet
import test_model_gpu as T
torch.set_num_threads(16)
arch, n_in, n_out, size, bs = "xresnet34", 4, 5, (64, 64), 2
torch.manual_seed(3)
ref = O.DynamicUnet(arch, n_in, n_out, size); O.randomize_bn_and_zero_gammas(ref, seed=4); T._make_bimodal(ref)
import copy
ref64 = copy.deepcopy(ref).double()
model = HipDynamicUnet(arch, n_in, n_out, size); model.load_state_dict(ref.state_dict())
x, y = O.synthetic_batch(bs, n_in, size[0], size[1], n_out)
w = torch.rand(n_out) + 0.5
ref.train(); model.train(); ref64.train()
O.CrossEntropyLossFlat(weight=w)(ref(x), y).backward()
O.CrossEntropyLossFlat(weight=w.double())(ref64(x.double()), y).backward()
model.forward_loss_backward(x.cuda(), y.cuda(), w.cuda()); torch.cuda.synchronize()
for (n, p), (_, q), (_, r) in zip(model.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
    s = r.grad.abs().max().item()
    if s == 0: continue
    eh = (p.grad.cpu().double() - r.grad).abs().max().item() / s
    ec = (q.grad.double() - r.grad).abs().max().item() / s
    if eh > 2e-3 or ec > 2e-3:
        print(f"{n:42s} scale {s:.3e} e_hip {eh:.3e} e_cpu {ec:.3e}  hip {p.grad.flatten()[:4].cpu().tolist()} f64 {r.grad.flatten()[:4].tolist()}")
