"""predict at batch 1 (the reference's per-tile loop, predict.py:191-193): eager launch stream against a replayed hipGraph, one process"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench as B
from unet_amd.model import HipDynamicUnet
dev = torch.device("cuda", 0)
for dt in ("f32", "bf16"):
    torch.manual_seed(0)
    m = HipDynamicUnet(B.ARCH, B.N_IN, B.N_CLS, (B.SIZE, B.SIZE), device=dev, act_dtype=dt)
    m.eval()
    x, _ = B.synth(1, 1, dev)
    for name, fn in (("eager", m.predict_probs), ("graph", m.predict_probs_graphed), ("eager", m.predict_probs), ("graph", m.predict_probs_graphed)):
        for _ in range(5):
            fn(x)
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(40):
                fn(x)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 40)
        print(f"{dt} batch 1 {name}: {1 / sorted(reps)[2]:8.1f} tiles/s  ({sorted(reps)[2] * 1e3:.3f} ms)", flush=True)
    p0, a0 = m.predict_probs(x); p0, a0 = p0.clone(), a0.clone()
    p1, a1 = m.predict_probs_graphed(x)
    torch.cuda.synchronize()
    print(dt, "graph == eager:", bool(torch.equal(p0, p1) and torch.equal(a0, a1)))
    del m
    torch.cuda.empty_cache()
