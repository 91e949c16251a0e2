"""Secondary measurements (not the headline): self-attention on, eval/predict throughput, cfg1 and cfg4 steps."""
import os, sys, time, json, torch
sys.path.insert(0, '.')
DT = os.environ.get("UNET_DTYPE", "f32")        # f32 | bf16 storage
from unet_amd.model import HipDynamicUnet
from unet_amd.optimizer import FlatAdam
from unet_amd.trainer import TrainStep


def synth(b, c, s, ncls, seed=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(0, 256, (b, c, s, s), generator=g).float() / 255).cuda(), torch.randint(0, ncls, (b, s, s), generator=g).cuda()


def train_rate(arch, c, ncls, s, b, sa=False, steps=4, graph=False):
    torch.manual_seed(0)
    m = HipDynamicUnet(arch, c, ncls, (s, s), self_attention=sa, act_dtype=DT); m.train()
    opt = FlatAdam(m, [1e-5, 3e-5, 1e-4]); st = TrainStep(m, opt, torch.full((ncls,), 1.0 / ncls, device='cuda'), use_graph=graph)
    x, y = synth(b, c, s, ncls)
    for _ in range(4): st(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    r = {"dtype": DT, "what": f"train {arch} {c}ch {s}x{s} {ncls}cls B={b} sa={sa} graph={graph}", "ms_per_step": round(dt * 1e3, 2), "tiles_per_s": round(b / dt, 2),
         "mem_GB": round(m.memory_bytes() / 2**30, 2)}
    print(json.dumps(r), flush=True)
    del m, opt, st
    torch.cuda.empty_cache()


def predict_rate(arch, c, ncls, s, b, steps=5, graph=False):
    torch.manual_seed(0)
    m = HipDynamicUnet(arch, c, ncls, (s, s), act_dtype=DT); m.eval()
    x, _ = synth(b, c, s, ncls)
    f = m.predict_probs_graphed if graph else m.predict_probs
    for _ in range(3): f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): f(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"dtype": DT, "what": f"predict {arch} {c}ch {s}x{s} B={b} graph={graph} (eval fwd + softmax + argmax)", "ms_per_batch": round(dt * 1e3, 2),
                      "tiles_per_s": round(b / dt, 2), "fwd_TFLOPs": round(b / dt * 255.846 / 1e3, 1) if arch == 'xresnet34' and s == 512 else None}), flush=True)
    del m
    torch.cuda.empty_cache()


which = sys.argv[1:] or ["sa", "predict", "cfg1", "cfg4"]
if "sa" in which: train_rate("xresnet34", 4, 5, 512, 16, sa=True)
if "predict" in which:
    predict_rate("xresnet34", 4, 5, 512, 16); predict_rate("xresnet34", 4, 5, 512, 1)
if "cfg1" in which:
    train_rate("xresnet18", 3, 2, 256, 2, steps=20); train_rate("xresnet18", 3, 2, 256, 2, steps=20, graph=True)
if "graph" in which:
    predict_rate("xresnet34", 4, 5, 512, 1, steps=20); predict_rate("xresnet34", 4, 5, 512, 1, steps=20, graph=True)
    train_rate("xresnet34", 4, 5, 512, 16, graph=True)
if "cfg4" in which: train_rate("xresnet50", 8, 10, 1024, 2, steps=2)


def cfg5_rate(side=20000, size=512, overlap=0.2, b=16):
    """BASELINE configs[4] on ONE GPU: sliding-window inference over a side x side 4-band raster resident in HBM as uint8
    (window rule of create_tiles_unet.split_raster: step = size - floor(size * overlap), last window flush with the border),
    batched eval forward + softmax, overlap merge (sum of probabilities + hit counter) and argmax on the device."""
    from unet_amd import ops
    torch.manual_seed(0)
    m = HipDynamicUnet("xresnet34", 4, 5, (size, size), act_dtype=DT); m.eval()
    g = torch.Generator(device="cuda").manual_seed(3)
    raster = torch.randint(0, 256, (4, side, side), dtype=torch.uint8, device="cuda", generator=g)
    step = size - int(size * overlap)
    offs = list(range(0, side - size + 1, step))
    if offs[-1] != side - size:
        offs.append(side - size)
    wins = [(y, x) for y in offs for x in offs]
    mosaic = torch.zeros((5, side, side), dtype=torch.float32, device="cuda")
    count = torch.zeros((side, side), dtype=torch.int32, device="cuda")
    am = torch.empty((side, side), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        m.predict_probs(torch.zeros(b, 4, size, size, device="cuda"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(0, len(wins), b):
        ws = wins[i:i + b]
        x = torch.stack([raster[:, y:y + size, x0:x0 + size] for y, x0 in ws]).float() / 255.0
        probs, _ = m.predict_probs(x, want_argmax=False)
        for j, (y, x0) in enumerate(ws):
            ops.mosaic_accumulate(probs[j], mosaic, count, y, x0)
    ops.mosaic_finalize(mosaic, count, am)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"dtype": DT, "what": f"cfg5 sliding-window predict {side}x{side}, {len(wins)} windows of {size}, overlap {overlap}, batch {b}, "
                              "forward + softmax + device merge + argmax", "seconds": round(dt, 2), "tiles_per_s": round(len(wins) / dt, 1),
                      "covered": bool((count > 0).all().item()), "max_overlap": int(count.max().item())}), flush=True)


if "cfg5" in which: cfg5_rate()

if "default" in which:     # the reference's shipped configuration (params_and_main.py): 3-band 400x400 tiles, 3 classes, batch 4, SA on
    train_rate("xresnet34", 3, 3, 400, 4, sa=True, steps=10)
    train_rate("xresnet34", 3, 3, 400, 16, sa=True, steps=5)
