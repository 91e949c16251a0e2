#!/usr/bin/env python
"""Headline benchmark: 512x512 tiles/s, forward + backward + optimizer step, 4-channel -> 5-class xresnet34
DynamicUnet (BASELINE.json configs[1]: batch 16 per MI355X; configs[2]: the same per GPU, tile-DDP over N GPUs).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `value` = tiles processed by all ranks / wall time of the K timed steps (max over
ranks), inputs resident in HBM.  `roofline` is the dominant kernel (the fp32 form of the 256-pixel implicit-GEMM conv tile,
its large-layer launches: forward and input-gradient of every wide 3x3 layer): summed algorithmic FLOPs / summed launch durations measured with
events on the launch stream inside the timed region; `roofline.traffic` comes from the committed counter summary it names
(`traffic_source`, `traffic_code`, `traffic_code_current`).  `cpu_baseline` = the torch-CPU oracle of the same step on this host's cores, on a
bounded sample (rank 0, N=1 only).  The N=1 run also carries `secondary`, every entry in its own try: the bf16-storage variant of configs[1]
(value un-probed, roofline from a second, probed run), predict at batch 16 / 1, `Learner.predict` per tile, configs[0] (cfg1), configs[3] (cfg4: xresnet50 8 -> 10,
1024 x 1024, fp32), the step with self-attention on (the reference's shipped default), configs[4] (cfg5: predict.predict_raster over a
20000 x 20000 raster) in fp32 and bf16 storage, `predict_files`: predict.save_predictions(merge=True) over 400 tile files, and `fit_files`: Learner.fit_one_cycle over tile FILES
through the product loader next to the resident-batch rate.  N > 1 runs add cfg5 over all ranks behind a watchdog that prints the headline with the failure
recorded and exits non-zero.  Every rank reports its own clock and the time its compute stream waited for the gradient all-reduce in `devices`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

GFLOP_PER_TILE_FWD_BWD = 767.388      # BASELINE.md section 2 (conv MACs x 2, cfg2)
GFLOP_PER_TILE_FWD = 255.846          # forward only (predict, cfg5)
PEAK_F32_TFLOPS = 157.3               # MI355X_MICROARCH.md: fp32 matrix == vector peak
PEAK_BF16_TFLOPS = 2500.0             # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; 6.3 TB/s achievable by a streaming copy)
ARCH, N_IN, N_CLS, SIZE = "xresnet34", 4, 5, 512


def synth(batch, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (batch, N_IN, SIZE, SIZE), generator=g).float() / 255
    y = torch.randint(0, N_CLS, (batch, SIZE, SIZE), generator=g)
    return x.to(device), y.to(device)


def _cpu_step_rate(O, arch, n_in, n_cls, size, tiles, iters, cores):
    torch.manual_seed(0)
    m = O.DynamicUnet(arch, n_in, n_cls, (size, size))
    m.train()
    opt = O.FastaiAdam(O.xresnet_split(m), list(O.even_mults(1e-5, 1e-4, 3)), no_wd=O.bn_bias_params(m))
    loss_fn = O.CrossEntropyLossFlat(weight=torch.full((n_cls,), 1.0 / n_cls))
    g = torch.Generator().manual_seed(1234)
    x = torch.randint(0, 256, (tiles, n_in, size, size), generator=g).float() / 255
    y = torch.randint(0, n_cls, (tiles, size, size), generator=g)

    def step():
        opt.zero_grad()
        loss_fn(m(x), y).backward()
        opt.step()

    t0 = time.perf_counter(); step()  # warm-up
    print(f"[bench] cpu_baseline {arch} {n_in}x{size}x{size} B={tiles}: warm-up {time.perf_counter() - t0:.1f}s on {cores} threads",
          file=sys.stderr, flush=True)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline iter {ts[-1]:.1f}s", file=sys.stderr, flush=True)
    return tiles / sorted(ts)[len(ts) // 2]


def cpu_baseline(sample_tiles: int = 2, iters: int = 3):
    """The reference's path restated on PyTorch-CPU (oracle/unet_oracle.py), same step, bounded sample (1 warm-up
    + 3 timed iterations; cfg2 geometry for the like-for-like tiles/s, plus BASELINE configs[0] = cfg1 at its batch 2)."""
    from oracle import unet_oracle as O
    # the GPU box gives one GPU's job a 16-core share; os.cpu_count() reports the whole host and would
    # oversubscribe oneDNN by an order of magnitude
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("UNET_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    rate = _cpu_step_rate(O, ARCH, N_IN, N_CLS, SIZE, sample_tiles, iters, cores)
    cfg1 = _cpu_step_rate(O, "xresnet18", 3, 2, 256, 2, iters, cores)
    return {"value": round(rate, 4), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample_tiles} tile(s) of the same 4x512x512 xresnet34 step (fwd + weighted CE + bwd + fastai-Adam), fp32, "
                      f"1 warm-up + {iters} timed, median",
            "cfg1_value": round(cfg1, 4),
            "cfg1_sample": f"BASELINE configs[0]: xresnet18 3->2, 256x256, batch 2, same step, 1 warm-up + {iters} timed, median (256x256 tiles/s)"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (the contract's env: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*) BEFORE this process has touched the GPU, wait for them, return the worst exit code.  No exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env))
    rc, live = 0, list(procs)
    while live:                                   # a rank that dies must not leave its peers waiting in a collective forever
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for q in live:
                    q.terminate()                 # exactly the children started above
    return rc


def step_bench(arch, n_in, n_cls, size, batch, dtype, steps, warmup, rank, world, dev, log, probe=True, use_graph=False, self_attention=False):
    """K timed fwd + weighted-CE + bwd + fastai-Adam steps on `batch` synthetic tiles per GPU.  Returns the measurements of this rank
    (wall time of the K steps bracketed by barrier + synchronize, dominant-kernel probe, time the compute stream waited for the gradient
    all-reduce)."""
    from unet_amd import ops as _ops
    from unet_amd.distributed import broadcast_parameters
    from unet_amd.model import HipDynamicUnet
    from unet_amd.optimizer import FlatAdam
    from unet_amd.trainer import TrainStep
    import torch.distributed as dist
    torch.manual_seed(0)
    model = HipDynamicUnet(arch, n_in, n_cls, (size, size), device=dev, act_dtype=dtype, self_attention=self_attention)
    broadcast_parameters(model.flat_param, list(model.buffers()))
    model.mark_weights_dirty()
    model.train()
    lr, enc_factor = 1e-4, 10.0        # params_and_main.py:53,84 defaults
    opt = FlatAdam(model, [lr / enc_factor, lr / enc_factor ** 0.5, lr])
    weights = torch.full((n_cls,), 1.0 / n_cls, device=dev)   # CLASS_WEIGHTS "even" (train.py:338-339)
    step = TrainStep(model, opt, weights, world, use_graph=use_graph)
    g = torch.Generator().manual_seed(1234 + rank)
    x = (torch.randint(0, 256, (batch, n_in, size, size), generator=g).float() / 255).to(dev)
    y = torch.randint(0, n_cls, (batch, size, size), generator=g).to(dev)
    log(f"{arch} {n_in}x{size}x{size} -> {n_cls} classes, {dtype}{', self-attention' if self_attention else ''}: {sum(p.numel() for p in model.parameters())} params, batch {batch}/gpu, world {world}")
    for i in range(warmup):
        step(x, y)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # both storage types: conv_bf16_t256_kernel (the 256-pixel tile of the wide 3x3 layers; variant ...7 = its large-layer launches), in its float or bf16 form
    pr = None
    if probe:
        _ops.CONV_PROBE = pr = _ops.ConvProbe(32 * 10000 + 128 * 10 + 7)
        pr.exclusive_of = model.ctx._side          # (weight gradients may run on a second stream: the probed launches are timed alone)
    step.comm_events = [] if world > 1 else None
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, y)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _ops.CONV_PROBE = None
    wait_ms = None
    if step.comm_events:
        wait_ms = sum(a.elapsed_time(b) for a, b in step.comm_events) / steps
    out = {"dt": dt, "t_local": t_local, "loss": float(loss.item()), "probe": None if pr is None else pr.summary(),
           "mem": model.memory_bytes(), "allreduce_wait_ms": wait_ms}
    del step, opt, model
    torch.cuda.empty_cache()
    return out


def roofline_of(ps, dtype, dt, steps, pmc):
    """dominant kernel over ALL its launches in the timed region: algorithmic FLOPs (fp32: MFMA roofline) or algorithmic bytes (bf16
    storage: HBM roofline) / summed launch durations (events on the launch stream)"""
    def traffic_of(name):
        """HBM bytes per launch of the kernel from the counter passes (profiles/pmc_traffic.json), per launch AS COUNTED HERE: a conv call
        that the library issues as two kernel launches (a narrow last channel block, a split reduction) is one launch of the probe"""
        t = pmc.get(name)
        if not t or not t.get("hbm_bytes_per_launch"):
            return None
        # where the figure comes from: it is read from a committed counter summary, not measured in this run; `traffic_code` is the
        # kernel-source hash that profile was taken on, `traffic_code_current` says whether the code running now is that code
        src["traffic_source"], src["traffic_code"] = t.get("source"), t.get("code")
        try:
            from unet_amd.build import source_hash
            src["traffic_code_current"] = (t.get("code") == source_hash()) if t.get("code") else None
        except Exception:      # noqa: BLE001
            src["traffic_code_current"] = None
        per_step = t.get("launches_sampled", 0) / max(1, t.get("steps_sampled", 0)) if t.get("steps_sampled") else None
        mine = ps["launches"] / max(1, steps)
        return int(t["hbm_bytes_per_launch"] * (per_step / mine if per_step and mine else 1.0))

    src = {"traffic_source": None, "traffic_code": None, "traffic_code_current": None}
    common = {"launches_per_step": ps["launches"] // max(1, steps), "avg_launch_ms": round(ps["avg_ms"], 4),
              "avg_launch_gflop": round(ps["flops"] / max(1, ps["launches"]) / 1e9, 2),
              "avg_launch_alg_bytes": int(ps["bytes"] / max(1, ps["launches"])),
              "share_of_step_time": round(ps["total_ms"] / (dt * 1e3), 4)}
    if dtype == "f32":
        achieved = ps["flops"] / (ps["total_ms"] * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "conv_bf16_t256_kernel<NTOT, 32, float> (the fp32 form of the 256-pixel x 128-channel implicit-GEMM tile, "
                                           "v_mfma_f32_16x16x4_f32: every wide 3x3 conv of the step, forward and input-gradient, incl. the 100-channel "
                                           "layers -- 7 channel tiles, transposed reduction tail; rocprofv3 lists its instantiations <6..8, 32, float>)",
                "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4),
                "traffic": traffic_of("conv_bf16_t256_kernel<float>"), **src, **common}
    achieved = ps["bytes"] / (ps["total_ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "conv_bf16_t256_kernel (bf16 storage, v_mfma_f32_16x16x32_bf16 implicit GEMM, 256-pixel x 128-channel tile: "
                                      "every wide 3x3 conv, forward and input-gradient; rocprofv3 lists its instantiations <6>, <7>, <8> = "
                                      "16-wide channel tiles per block); algorithmic bytes = every operand tensor once in, the result once "
                                      "out, the packed filter once",
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4),
            "traffic": traffic_of("conv_bf16_t256_kernel"), **src,
            "mfma_tflops": round(ps["flops"] / (ps["total_ms"] * 1e-3) / 1e12, 1),
            "mfma_frac_of_bf16_peak": round(ps["flops"] / (ps["total_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), **common}


def predict_bench(dtype, batch, dev, iters=6):
    """eval forward + softmax + argmax (what learn.predict computes per tile, predict.py:193-203,232) on `batch` cfg2 tiles"""
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    m = HipDynamicUnet(ARCH, N_IN, N_CLS, (SIZE, SIZE), device=dev, act_dtype=dtype)
    m.eval()
    x, _ = synth(batch, 1, dev)
    for _ in range(3):
        m.predict_probs(x)
    torch.cuda.synchronize()
    reps = []          # median of three repetitions: a batch-1 window is 30 ms long, one host hiccup inside it halves the figure
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters):
            m.predict_probs(x)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / iters)
    dt = sorted(reps)[1]
    del m
    torch.cuda.empty_cache()
    return {"value": round(batch / dt, 1), "unit": "tiles/s", "ms_per_batch": round(dt * 1e3, 3), "dtype": dtype, "batch": batch,
            "fwd_tflops": round(batch / dt * GFLOP_PER_TILE_FWD / 1e3, 1)}


def learner_predict_bench(dtype, dev, iters=20):
    """the reference's literal prediction loop (predict.py:191-193: `learn.predict(tile)` per tile): Learner.predict on one uint8 4x512x512
    tile -- integer upload, scaling on the device, eval forward + softmax + argmax, class probabilities and mask back on the host"""
    import numpy as np
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    m = HipDynamicUnet(ARCH, N_IN, N_CLS, (SIZE, SIZE), device=dev, act_dtype=dtype)
    dls = DataLoaders(TileDataset([np.zeros((N_IN, SIZE, SIZE), np.uint8)], None, "int8"), None, 16, device=dev, vocab=list("abcde"))
    ln = Learner(dls, m, loss_func=CrossEntropyLossFlat(axis=1))
    x = np.random.default_rng(0).integers(0, 256, (N_IN, SIZE, SIZE)).astype(np.uint8)
    for _ in range(3):
        ln.predict(x)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters):
            out = ln.predict(x)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / iters)
    dt = sorted(reps)[1]
    r = {"value": round(1.0 / dt, 1), "unit": "tiles/s", "ms_per_tile": round(dt * 1e3, 3), "dtype": dtype,
         "returns": [list(out[1].shape), list(out[2].shape)]}
    del ln, m
    torch.cuda.empty_cache()
    return r


def cfg5_bench(dtype, dev, side=20000, size=512, overlap=0.2, batch=16):
    """BASELINE configs[4] on this GPU through the product entry point predict.predict_raster: uint8 raster resident in HBM, windows
    cut + scaled on the device, batched forward, softmax + overlap merge in window order, argmax; only the uint8 mask reaches the host"""
    import predict as P
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    m = HipDynamicUnet(ARCH, N_IN, N_CLS, (size, size), device=dev, act_dtype=dtype)
    m.eval()
    g = torch.Generator(device=dev).manual_seed(3)
    raster = torch.randint(1, 256, (N_IN, side, side), dtype=torch.uint8, device=dev, generator=g)
    P.predict_raster(m, raster[:, :2 * size, :4 * size].contiguous(), size, overlap, batch_size=batch)      # warm-up: buffers, packed filters
    tm = {}
    out = P.predict_raster(m, raster, size, overlap, batch_size=batch, timing=tm)
    r = {"value": round(tm["windows"] / tm["seconds"], 1), "unit": "tiles/s", "seconds": round(tm["seconds"], 3), "windows": tm["windows"],
         "dtype": dtype, "raster": [N_IN, side, side], "window": size, "overlap": overlap, "batch": batch,
         "hits_min": tm["hits_min"], "hits_max": tm["hits_max"], "mask_shape": list(out.shape), "mask_checksum": int(out.astype("int64").sum()),
         "fwd_tflops": round(tm["windows"] / tm["seconds"] * GFLOP_PER_TILE_FWD / 1e3, 1),
         "what": "predict.predict_raster end to end incl. the 400 MB uint8 mask device -> host"}
    del m, raster, out
    torch.cuda.empty_cache()
    return r


def _write_tile_files(root: Path, n: int, log):
    """n uint8 4x512x512 image tiles + uint8 masks as GeoTIFFs, once uncompressed (unet_amd.tiffio.write_tiff: what the reference's
    create_tiles_unet.save_crop writes through GDAL's default creation options), once LZW-compressed and once JPEG-compressed (quality 90,
    masks LZW) by libtiff (Pillow; both skipped when Pillow is missing).  Content: a ramp + low-amplitude noise, so that LZW compresses to ~2/3 like imagery does (uniform noise would make
    every code a literal)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from unet_amd.tiffio import write_tiff
    try:
        from PIL import Image
    except Exception:      # noqa: BLE001
        Image = None
    sets = {"none": root / "none", "lzw": root / "lzw", "jpeg": root / "jpeg"} if Image is not None else {"none": root / "none"}
    for d in sets.values():
        (d / "img").mkdir(parents=True)
        (d / "mask").mkdir(parents=True)
    ramp = (np.add.outer(np.arange(SIZE), np.arange(SIZE)) // 8)[None] + (np.arange(N_IN) * 7)[:, None, None]

    def one(i):
        g = np.random.default_rng(1000 + i)
        img = ((ramp + 31 * i) + g.integers(0, 3, (N_IN, SIZE, SIZE))).astype(np.uint8)
        blocks = g.integers(0, N_CLS, (SIZE // 32, SIZE // 32), dtype=np.uint8)
        mask = np.repeat(np.repeat(blocks, 32, 0), 32, 1)
        gt = (1000.0 + 100 * i, 0.2, 0.0, 5000.0, 0.0, -0.2)
        write_tiff(sets["none"] / "img" / f"t{i:04d}.tif", img, geotransform=gt)
        write_tiff(sets["none"] / "mask" / f"t{i:04d}.tif", mask, geotransform=gt)
        if "lzw" in sets:
            Image.fromarray(np.moveaxis(img, 0, -1), "RGBA").save(sets["lzw"] / "img" / f"t{i:04d}.tif", compression="tiff_lzw")
            Image.fromarray(mask, "L").save(sets["lzw"] / "mask" / f"t{i:04d}.tif", compression="tiff_lzw")
        if "jpeg" in sets:          # COMPRESS=JPEG image tiles (lossy: quality 90), masks lossless (LZW) as they have to be
            Image.fromarray(np.moveaxis(img, 0, -1), "RGBA").save(sets["jpeg"] / "img" / f"t{i:04d}.tif", compression="jpeg", quality=90)
            Image.fromarray(mask, "L").save(sets["jpeg"] / "mask" / f"t{i:04d}.tif", compression="tiff_lzw")

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(one, range(n)))
    log(f"fit_files: wrote {n} tiles x {len(sets)} encodings in {time.perf_counter() - t0:.1f} s under {root}")
    return sets


def fit_files_bench(dev, log, resident: dict, n_tiles: int = 512, batch: int = 16):
    """The workflow the reference actually runs (train.py:345 -> learn.fit_one_cycle, data.py:18-28, utils.py:239-295): ONE epoch of
    Learner.fit_one_cycle over tile FILES through the product loader (unet_amd/feed.py: decode pool -> pinned integer staging -> asynchronous
    upload -> scaling / mask widening / flips on the device), default flip augmentation on, timed end to end (first file open to the last
    optimizer step) and put next to the resident-batch step rate of the same process."""
    import shutil
    import tempfile
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, FlipAugment, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    root = Path(tempfile.mkdtemp(prefix="unet_fit_files_"))
    out = {"tiles": n_tiles, "batch": batch, "what": "Learner.fit_one_cycle(1) over uint8 4x512x512 GeoTIFF tile files + uint8 masks, device feed "
                                                       "(unet_amd/feed.py), flips on (n_transform_imgs 0.5); tiles/s end to end; ratio = / the "
                                                       "resident-batch step rate measured in this process"}
    try:
        sets = _write_tile_files(root, n_tiles, log)
        for enc, d in sets.items():
            imgs = sorted((d / "img").iterdir())
            masks = [d / "mask" / p.name for p in imgs]
            for dtype in ("f32", "bf16"):
                torch.manual_seed(0)
                model = HipDynamicUnet(ARCH, N_IN, N_CLS, (SIZE, SIZE), device=dev, act_dtype=dtype)
                w = torch.full((N_CLS,), 1.0 / N_CLS, device=dev)

                def learner(k):
                    dls = DataLoaders(TileDataset(imgs[:k], masks[:k], "int8"), None, batch, device=dev, vocab=list("abcde"),
                                      train_tfm=FlipAugment(n_transform_imgs=0.5, seed=1))
                    ln = Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1, weight=w), path=root)
                    ln._no_logging = True
                    return ln
                learner(4 * batch).fit_one_cycle(1, lr_max=slice(1e-5, 1e-4))        # warm-up: buffers, packed filters, page cache of the first files
                ln = learner(n_tiles)
                for _ in zip(range(2), ln.dls.train):                                # the loader's threads and pinned ring exist before the clock starts
                    pass
                for p in imgs + masks:                                              # page cache warm (the box's disk is not the subject)
                    with open(p, "rb") as f:
                        f.read()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ln.fit_one_cycle(1, lr_max=slice(1e-5, 1e-4))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                steps = n_tiles // batch
                v = steps * batch / dt
                res = resident.get(dtype)
                out[f"{enc}_{dtype}"] = {"value": round(v, 2), "unit": "tiles/s", "seconds": round(dt, 3), "steps": steps,
                                         "ms_per_step": round(dt / steps * 1e3, 3), "resident": res,
                                         "ratio_to_resident": None if not res else round(v / res, 4),
                                         "final_train_loss": round(float(ln.recorder.losses[-1]), 5),
                                         "feed_workers": ln.dls.train._feeder.workers if ln.dls.train._feeder is not None else None}
                log(f"fit_files {enc} {dtype}: {v:.1f} tiles/s ({dt:.2f} s), resident {res}")
                del ln, model
                torch.cuda.empty_cache()
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def predict_files_bench(dev, log, side: int = 8000):
    """The reference's prediction entry point on tile FILES (predict.py:146-355): an 8000 x 8000 uint8 scene is cut by
    create_tiles_unet.split_raster into 400 GeoTIFF tiles of 512 with overlap 0.2, an exported model is loaded by
    predict.save_predictions(merge=True), which reads, predicts, overlap-merges and writes the mask.  Timed end to end (model load to mask on
    disk) and inside the engine (first tile read to merged mask), fp32 and bf16 storage; the mask must equal predict_raster's on the scene."""
    import shutil
    import tempfile
    import numpy as np
    import create_tiles_unet as T
    import predict as P
    from unet_amd.learner import CrossEntropyLossFlat, DataLoaders, Learner, TileDataset
    from unet_amd.model import HipDynamicUnet
    from unet_amd.tiffio import read_tiff, write_tiff
    root = Path(tempfile.mkdtemp(prefix="unet_predict_files_"))
    out = {"what": "create_tiles_unet.split_raster -> 400 tile files -> predict.save_predictions(merge=True), end to end incl. load_learner and the "
                   "mask GeoTIFF; engine = first tile read to merged mask", "raster": [N_IN, side, side], "batch": 16}
    try:
        g = np.random.default_rng(0)
        img = g.integers(1, 256, (N_IN, side, side), dtype=np.uint8)
        write_tiff(root / "scene.tif", img, geotransform=(400000.0, 0.5, 0.0, 5700000.0, 0.0, -0.5))
        n = T.split_raster(str(root / "scene.tif"), None, str(root / "cut"), patch_size=SIZE, patch_overlap=0.2, split=[1])["tiles"]
        tiles_dir = root / "cut" / "img_tiles"
        for dtype in ("f32", "bf16"):
            torch.manual_seed(0)
            model = HipDynamicUnet(ARCH, N_IN, N_CLS, (SIZE, SIZE), device=dev, act_dtype=dtype)
            dls = DataLoaders(TileDataset([np.zeros((N_IN, SIZE, SIZE), np.uint8)], None, "int8"), None, 1, device=dev, vocab=list("abcde"))
            pkl = root / f"m_{dtype}.pkl"
            Learner(dls, model, loss_func=CrossEntropyLossFlat(axis=1), path=root).export(pkl)
            model.eval()
            ref = P.predict_raster(model, img, SIZE, 0.2, batch_size=16)
            del model
            torch.cuda.empty_cache()
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):          # (save_predictions prints progress lines: the bench prints ONE JSON line)
                P.save_predictions(str(pkl), str(tiles_dir), False, merge=True, AOI="warm", validation_vision=False, batch_size=16)
                tm = {}
                t0 = time.perf_counter()
                f = P.save_predictions(str(pkl), str(tiles_dir), False, merge=True, AOI=f"timed_{dtype}", validation_vision=False, batch_size=16, timing=tm)
                dt = time.perf_counter() - t0
            same = bool(np.array_equal(read_tiff(f)[0], ref))
            out[dtype] = {"value": round(n / dt, 1), "unit": "tiles/s", "seconds": round(dt, 3), "tiles": n,
                          "engine_tiles_per_s": round(n / tm["seconds"], 1), "mask_equals_predict_raster": same}
            log(f"predict_files {dtype}: {n / dt:.1f} tiles/s end to end, engine {n / tm['seconds']:.1f}, mask equal {same}")
            torch.cuda.empty_cache()
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def cfg5_multi(dtype, dev, rank, world, side=int(os.environ.get("UNET_CFG5_SIDE", "20000")), size=512, overlap=0.2, batch=16):
    """BASELINE configs[4] as worded: the 20000 x 20000 raster predicted by ALL ranks -- predict.predict_raster partitions the windows into
    contiguous row blocks (one per rank), every rank keeps its strip of the mosaic, overlap rows travel as slabs to the neighbouring rank
    (RCCL send / recv over xGMI), rank 0 gathers the uint8 mask.  Wall time between two barriers, max over ranks."""
    import torch.distributed as dist
    import predict as P
    from unet_amd.distributed import broadcast_parameters
    from unet_amd.model import HipDynamicUnet
    torch.manual_seed(0)
    m = HipDynamicUnet(ARCH, N_IN, N_CLS, (size, size), device=dev, act_dtype=dtype)
    broadcast_parameters(m.flat_param, list(m.buffers()))
    m.mark_weights_dirty()
    m.eval()
    g = torch.Generator(device=dev).manual_seed(3)                      # the same raster on every rank (a real run reads one file)
    raster = torch.randint(1, 256, (N_IN, side, side), dtype=torch.uint8, device=dev, generator=g)
    P.predict_raster(m, raster[:, :4 * size, :4 * size].contiguous(), size, overlap, batch_size=batch)      # warm-up incl. the p2p channels
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = {}
    out = P.predict_raster(m, raster, size, overlap, batch_size=batch, timing=tm)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, tm["seconds"]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "windows": tm["windows_this_rank"], "seconds": round(tm["seconds"], 3),
                                      "strip_rows": tm["strip_rows"], "slab_MB_sent": round(tm["slab_floats_sent"] * 4 / 1e6, 1)})
    r = None
    if rank == 0:
        r = {"value": round(tm["windows"] / float(t[0]), 1), "unit": "tiles/s", "seconds": round(float(t[0]), 3), "windows": tm["windows"],
             "dtype": dtype, "n_gpus": world, "active_ranks": tm["active_ranks"], "raster": [N_IN, side, side], "window": size,
             "overlap": overlap, "batch": batch, "mask_shape": list(out.shape), "mask_checksum": int(out.astype("int64").sum()),
             "per_rank": per_rank, "what": "predict.predict_raster over all ranks: row-block partition, slab exchange, uint8 gather on rank 0"}
    del m, raster, out
    torch.cuda.empty_cache()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (bf16 step, predict, cfg1, cfg5) of the N=1 run")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 (default): the parity path, the headline.  bf16: bf16 storage / fp32 accumulate variant of configs[1]")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # this process never initialises the GPU
    if int(os.environ["WORLD_SIZE"] if "WORLD_SIZE" in os.environ else 1) != args.gpus:
        world = int(os.environ["WORLD_SIZE"])
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or run "
                         f"`python bench.py --gpus {args.gpus}` without WORLD_SIZE set and let it start the ranks itself")
    from unet_amd.distributed import init_from_env
    rank, local_rank, world = init_from_env()
    if os.environ.get("UNET_FORCE_DEVICE") is not None:      # rehearsal of the N>1 path on a single GPU (with UNET_DIST_BACKEND=gloo)
        local_rank = int(os.environ["UNET_FORCE_DEVICE"])
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    res = step_bench(ARCH, N_IN, N_CLS, SIZE, args.batch, args.dtype, args.steps, args.warmup, rank, world, dev, log)
    dt = res["dt"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ps = res["probe"]
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": f"cuda:{local_rank}", "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
          # this rank's own clock over the K steps (before the closing barrier) and the time per step its compute stream spent waiting
          # for gradient buckets that were still in flight when the backward ended: what a scaling curve needs to explain itself
          "ms_per_step_local": round(res["t_local"] / args.steps * 1e3, 3),
          "allreduce_wait_ms_per_step": None if res["allreduce_wait_ms"] is None else round(res["allreduce_wait_ms"], 3),
          "dominant_kernel_avg_ms": round(ps["avg_ms"], 4)}
    devices = [me]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)

    if rank == 0:
        tiles = args.batch * world * args.steps
        value = tiles / dt
        pmc = ROOT / "profiles" / "pmc_traffic.json"      # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
        pmc = json.loads(pmc.read_text()) if pmc.exists() else {}
        out = {
            "metric": "512x512 tiles/sec fwd+bwd (4-ch->5-class U-Net)", "value": round(value, 3), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg2: 4x512x512 tiles, xresnet34 DynamicUnet, 5 classes, fwd+bwd+fastai-Adam step, "
                                   "BN train mode, CE weights even" + ("" if args.dtype == "f32" else
                                   "; bf16 storage of activations / gradients / packed filters, fp32 accumulate, fp32 master weights + Adam"),
                       "tiles_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"tile-dp{world}", "self_attention": False},
            "loss": round(res["loss"], 5),
            "step_tflops": round(value * GFLOP_PER_TILE_FWD_BWD / 1e3, 2),
            "step_frac_of_f32_peak": round(value * GFLOP_PER_TILE_FWD_BWD / 1e3 / (PEAK_F32_TFLOPS * world), 4),
            "roofline": roofline_of(ps, args.dtype, dt, args.steps, pmc),
            "hbm_bytes_allocated": res["mem"],
            # world size the collective library itself reports (1 = no process group) and the device every rank ran on
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "dist_backend": (dist.get_backend() if world > 1 else None),
            "devices": devices,
        }
        if world == 1 and not args.no_secondary and args.dtype == "f32":
            # measured in this same process, after the headline: the other BASELINE configurations and the bf16-storage variant of configs[1].
            # Every entry runs in its own try: a failure there is recorded in the entry and the headline line is still printed.
            sec = {}

            def guarded(name, what, fn):
                log(f"secondary: {what}")
                try:
                    sec[name] = fn()
                except Exception as e:      # noqa: BLE001  (the headline must survive)
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
                    log(f"secondary {name} FAILED: {type(e).__name__}: {e}")
                    torch.cuda.empty_cache()

            def step_line(r, batch, steps, gflop, peak, workload, dtype, with_roofline=False):
                v = batch * steps / r["dt"]
                o = {"value": round(v, 3), "unit": "tiles/s", "ms_per_step": round(r["dt"] / steps * 1e3, 3), "steps": steps, "dtype": dtype,
                     "batch": batch, "workload": workload, "loss": round(r["loss"], 5), "step_tflops": round(v * gflop / 1e3, 2),
                     "step_frac_of_peak": round(v * gflop / 1e3 / peak, 4), "peak_tflops": peak, "hbm_bytes_allocated": r["mem"]}
                if with_roofline:
                    o["roofline"] = roofline_of(r["probe"], dtype, r["dt"], steps, pmc)
                return o

            def bf16_line():
                # `value`: K un-probed steps (bf16 storage runs its weight gradients on a second stream; the probe's events would make the
                # launch stream wait for that stream in front of every probed launch -- measured: 3-5 % of the step).  `roofline`: a second
                # run of the same step WITH the probe (launches timed alone), its own tiles/s reported as `probed_value`
                rb = step_bench(ARCH, N_IN, N_CLS, SIZE, args.batch, "bf16", args.steps, args.warmup, 0, 1, dev, log, probe=False)
                o = step_line(rb, args.batch, args.steps, GFLOP_PER_TILE_FWD_BWD, PEAK_BF16_TFLOPS,
                              "cfg2 (BASELINE configs[1] wording): same step, bf16 storage of activations / gradients / packed filters, "
                              "fp32 accumulate, fp32 master weights + Adam", "bf16")
                o["step_frac_of_bf16_peak"] = o["step_frac_of_peak"]
                rp = step_bench(ARCH, N_IN, N_CLS, SIZE, args.batch, "bf16", 5, 2, 0, 1, dev, log, probe=True)
                o["roofline"] = roofline_of(rp["probe"], "bf16", rp["dt"], 5, pmc)
                o["roofline"]["measured_in"] = "a second run of 5 steps with the launch probe on (probed launches wait for the weight-gradient stream: timed alone)"
                o["roofline"]["probed_value"] = round(args.batch * 5 / rp["dt"], 3)
                return o

            def sa_line(dtype):
                # the reference's SHIPPED default is self_attention=True (params_and_main.py:81-83); SURVEY 8(d): report it separately
                r = step_bench(ARCH, N_IN, N_CLS, SIZE, args.batch, dtype, 5, 2, 0, 1, dev, log, probe=False, self_attention=True)
                o = step_line(r, args.batch, 5, GFLOP_PER_TILE_FWD_BWD + 3 * 16.0, PEAK_F32_TFLOPS if dtype == "f32" else PEAK_BF16_TFLOPS,
                              "cfg2 with DynamicUnet(self_attention=True): SelfAttention(384) at 64x64 positions behind UnetBlock 1 "
                              "(+ ~16 GFLOP fwd per tile)", dtype)
                o["attention"] = ("fused kernels (csrc/attention.hip: no N x N tensor in HBM)" if dtype == "bf16"
                                  else "blockwise products on the conv / weight-gradient kernels (fp32 parity path)")
                return o

            def cfg4_line():
                # BASELINE configs[3]: 8-ch 1024x1024 tiles, xresnet50 encoder, 10 classes, fp32
                r = step_bench("xresnet50", 8, 10, 1024, 2, "f32", 3, 2, 0, 1, dev, log, probe=False)
                o = step_line(r, 2, 3, 30454.37, PEAK_F32_TFLOPS, "cfg4 (BASELINE configs[3]): xresnet50 DynamicUnet 8 -> 10 classes, "
                              "1024x1024 tiles, batch 2, fp32, same fwd + CE + bwd + fastai-Adam step", "f32")
                o["unit"] = "1024x1024 tiles/s"
                return o

            def cfg1_line():
                r1 = step_bench("xresnet18", 3, 2, 256, 2, "f32", 30, 5, 0, 1, dev, log, probe=False)
                return {"value": round(2 * 30 / r1["dt"], 2), "unit": "256x256 tiles/s", "ms_per_step": round(r1["dt"] / 30 * 1e3, 3), "dtype": "f32",
                        "step_tflops": round(2 * 30 / r1["dt"] * 175.805 / 1e3, 2)}

            guarded("bf16", "bf16-storage step", bf16_line)
            guarded("predict_b16", "predict (eval forward + softmax + argmax), batch 16",
                    lambda: {"f32": predict_bench("f32", 16, dev), "bf16": predict_bench("bf16", 16, dev)})
            guarded("predict_b1", "predict, batch 1",
                    lambda: {"f32": predict_bench("f32", 1, dev, iters=20), "bf16": predict_bench("bf16", 1, dev, iters=20)})
            guarded("learner_predict", "Learner.predict per tile (the reference's loop, predict.py:191-193)",
                    lambda: {"f32": learner_predict_bench("f32", dev), "bf16": learner_predict_bench("bf16", dev)})
            guarded("cfg1", "cfg1 (BASELINE configs[0]: xresnet18 3->2, 256x256, batch 2)", cfg1_line)
            guarded("cfg4", "cfg4 (BASELINE configs[3]: xresnet50 8->10, 1024x1024, batch 2, fp32)", cfg4_line)
            guarded("sa_on", "cfg2 with self-attention on (the reference's shipped default), fp32 + bf16",
                    lambda: {"f32": sa_line("f32"), "bf16": sa_line("bf16")})
            guarded("cfg5", "cfg5 (BASELINE configs[4]: 20000x20000 sliding-window predict) fp32 + bf16",
                    lambda: {"f32": cfg5_bench("f32", dev), "bf16": cfg5_bench("bf16", dev)})
            guarded("predict_files", "save_predictions(merge=True) over 400 tile files (split_raster -> files -> merged mask), fp32 + bf16",
                    lambda: predict_files_bench(dev, log))
            guarded("fit_files", "fit_one_cycle from tile files through the product loader (uncompressed + LZW + JPEG, fp32 + bf16)",
                    lambda: fit_files_bench(dev, log, {"f32": round(value, 3), "bf16": (sec.get("bf16") or {}).get("value")}))
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if world > 1 and not args.no_secondary and args.dtype == "f32":
        # configs[4] over all ranks, AFTER the headline has been measured.  The exchange uses RCCL point-to-point calls the training step
        # does not: a watchdog makes sure the headline line is printed even if this part should not come back
        import threading
        done = threading.Event()

        def emergency():
            # the exchange did not come back: print the headline WITH the failure recorded in it, then end this rank with a non-zero
            # code so that spawn_ranks / the launcher see a failed run (a hang must not read as success)
            if done.is_set():
                return
            if rank == 0:
                out["secondary"] = {"cfg5": {"error": "multi-rank cfg5 did not finish within 300 s (collective or GPU hang); headline measured before it"}}
                out["failed"] = "secondary.cfg5 watchdog"
                print(json.dumps(out), flush=True)
            os._exit(3)
        timer = threading.Timer(300.0, emergency)
        timer.daemon = True
        timer.start()
        sec = {}
        try:
            log("secondary: cfg5 over all ranks, fp32")
            r32 = cfg5_multi("f32", dev, rank, world)
            log("secondary: cfg5 over all ranks, bf16")
            r16 = cfg5_multi("bf16", dev, rank, world)
            sec = {"cfg5": {"f32": r32, "bf16": r16}}
        except Exception as e:      # noqa: BLE001  (the headline must survive)
            sec = {"cfg5": {"error": f"{type(e).__name__}: {e}"}}
        done.set()
        timer.cancel()
        if rank == 0:
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
