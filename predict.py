"""Prediction entry points of the tile workflow on the MI355X hot path.

``save_predictions`` mirrors the reference's (``predict.py:146-147``): a folder of tile files -> per-tile predictions or ONE merged
raster.  ``predict_raster`` is BASELINE.json configs[4] as one call: sliding-window inference over a whole raster that stays in HBM as
the integers it was read as -- the reference needs two steps for it, ``create_tiles_unet.split_raster`` (tile files on disk,
``create_tiles_unet.py:252-434``) and ``save_predictions(merge=True)`` (``predict.py:191-222,257-334``), and produces the same mosaic.

Differences that keep the results but not the schedule:
* tiles are predicted in BATCHES (the reference loops ``learn.predict`` one tile at a time, ``predict.py:191-193``); windows / tiles are cut,
  cast and scaled on the device (``unet_window_gather``) from uint8 / uint16 samples, a prefetch thread decodes the next tile files into pinned
  memory while the current batch runs;
* the overlap merge -- sum of softmax probabilities + hit counter -> divide -> argmax (``predict.py:284-334``) -- runs on the GPU, batch by
  batch (``unet_mosaic_accumulate_windows``), in ONE defined order (placements sorted by row, then column; the reference uses the directory
  order of ``glob``);
* under N ranks (one process per GPU) the mosaic is partitioned by rows (``unet_amd/mosaic.py``): every rank keeps only its strip, overlap rows
  travel as per-window slabs to the neighbouring rank, and only the requested band(s) -- the uint8 argmax by default -- reach rank 0 and the host.
``regression`` predicts the raw single-band output (merged mosaic = mean of the overlapping tiles, nodata -9999 where no tile was
placed).  The confusion-matrix plots (``predict.py:56-143``) are reporting and out of scope.
"""
from __future__ import annotations

import contextlib
import os
import queue
import threading
import time
import warnings
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from unet_amd import ops
from unet_amd.learner import load_learner, open_tile
from unet_amd.mosaic import MergePlan, keep_windows, merge_order, sliding_windows
from unet_amd.tiffio import read_tiff, tiff_info, write_tiff


def store_tif(output_file, data, geotrans=None, tags=None, nodata=None, class_zero=False):
    """predict.py:19-52: GeoTIFF writer; class_zero shifts class ids back by one (0 was reserved for nodata)."""
    a = np.asarray(data)
    if class_zero and a.dtype.kind in "ui":
        a = a - 1 if a.dtype.kind == "i" else (a.astype(np.int16) - 1)
    write_tiff(output_file, a, geotransform=geotrans, tags=tags, nodata=nodata)


def _geo(path):
    """(geotransform, GeoTIFF tags, height, width) from the header alone (.npy tiles carry no georeference)"""
    if Path(path).suffix == ".npy":
        a = np.load(path, mmap_mode="r")
        return None, {}, int(a.shape[-2]), int(a.shape[-1])
    meta = tiff_info(path)
    return meta["geotransform"], meta["tags"], meta["height"], meta["width"]


LARGE_FILE_SCALE = (128 / 4) - 1        # predict.py:209-214: probabilities stretched to int8 as around(p * 31)
_SAMPLE_TYPES = (np.uint8, np.uint16, np.int16, np.int32, np.float32)


def _as_samples(a: np.ndarray) -> np.ndarray:
    """sample array the device kernels read as is; anything else goes through int32, which is what data.py:24 does to every tile anyway"""
    return a if a.dtype.type in _SAMPLE_TYPES else a.astype(np.int32)


def _torch_samples(a: np.ndarray) -> torch.Tensor:
    a = np.ascontiguousarray(_as_samples(a))
    if a.dtype == np.uint16:        # torch.from_numpy has no uint16 before 2.3; view the bits
        return torch.from_numpy(a.view(np.int16)).view(torch.uint16)
    return torch.from_numpy(a)


# ----------------------------------------------------------------------------------------------- p2p plumbing (RCCL, or gloo in tests)

def _dist():
    import torch.distributed as dist
    return dist


def _exchange(send: Optional[torch.Tensor], dst: int, recv_numel: int, src: int, dtype, device) -> Optional[torch.Tensor]:
    """one simultaneous isend / irecv pair between neighbouring ranks (either side may be absent); gloo moves host memory"""
    dist = _dist()
    host = dist.get_backend() == "gloo"
    ops_, rbuf = [], None
    if recv_numel:
        rbuf = torch.empty(recv_numel, dtype=dtype, device="cpu" if host else device)
        ops_.append(dist.P2POp(dist.irecv, rbuf, src))
    if send is not None and send.numel():
        ops_.append(dist.P2POp(dist.isend, send.cpu() if host else send, dst))
    if ops_:
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
    return None if rbuf is None else rbuf.to(device)


# ----------------------------------------------------------------------------------------------- the merge engine

class _Merge:
    """Runs one rank's share of a merged prediction: forward of its placements in batches, accumulation into its strip of the
    mosaic, slab exchange with the neighbours, finalisation, gather of the requested output on rank 0."""

    def __init__(self, model, places: np.ndarray, MH: int, MW: int, regression: bool, int8_merge: bool, rank: int, world: int, batch: int):
        self.model, self.dev = model, model._device
        self.C = model.n_out
        self.raw, self.int8 = bool(regression), bool(int8_merge)
        self.rank, self.world, self.batch = rank, world, batch
        self.plan = MergePlan(places, MH, MW, world)
        self.lo, self.hi = self.plan.own[rank]
        rows = self.hi - self.lo
        if self.int8:       # the reference's int8 arrays: merged raster AND hit counter are int8 per class (predict.py:276-281)
            self.mosaic = torch.zeros((self.C, max(rows, 1), MW), dtype=torch.int8, device=self.dev)
            self.count = torch.zeros((self.C, max(rows, 1), MW), dtype=torch.int8, device=self.dev)
        else:
            self.mosaic = torch.zeros((self.C, max(rows, 1), MW), dtype=torch.float32, device=self.dev)
            self.count = torch.zeros((max(rows, 1), MW), dtype=torch.int32, device=self.dev)
        self.acc_table = ops.window_table(self.plan.places[:, :2].tolist(), self.dev)       # (y0, x0) in mosaic coordinates
        self.my_slabs = self.plan.slabs(rank)
        self._slab_off, off = {}, 0
        for i, r in self.my_slabs:
            self._slab_off[i] = (off, r)
            off += self.C * r * int(self.plan.places[i, 3])
        self.sendbuf = torch.empty(off, dtype=torch.float32, device=self.dev) if off else None

    # -- int8 "large_file" accumulation of one window's probabilities [C, rows, w] at strip row y (may be clipped), column x
    def _add_int8(self, probs: torch.Tensor, y: int, x: int):
        q = torch.round(probs * LARGE_FILE_SCALE).to(torch.int8)            # np.around: half to even, as torch.round
        r0, r1 = max(0, -y), min(q.shape[1], self.hi - self.lo - y)
        if r1 <= r0:
            return
        self.mosaic[:, y + r0:y + r1, x:x + q.shape[2]] += q[:, r0:r1]
        self.count[:, y + r0:y + r1, x:x + q.shape[2]] += 1

    def add_batch(self, first: int, n: int, z: ops.TS):
        """logits z [>= n, h, w, C] of placements [first, first + n)"""
        rows = self.hi - self.lo
        if rows > 0:
            if self.int8:
                probs = torch.empty((n, self.C, z.H, z.W), dtype=torch.float32, device=self.dev)
                ops.softmax_argmax(ops.TS(z.buf[:n], z.co, z.C), probs, None)
                for j in range(n):
                    y0, x0 = self.plan.places[first + j, :2]
                    self._add_int8(probs[j], int(y0) - self.lo, int(x0))
            else:
                ops.mosaic_accumulate_windows(z, self.acc_table, first, n, (self.lo, 0), self.mosaic, self.count, 0, rows, raw=self.raw)
        for j in range(n):          # rows that belong to the strip above: per-window slabs for rank - 1
            ent = self._slab_off.get(first + j)
            if ent is None:
                continue
            off, r = ent
            w = int(self.plan.places[first + j, 3])
            out = self.sendbuf[off:off + self.C * r * w].view(1, self.C, r, w)
            zs = ops.TS(z.buf[j:j + 1, :r], z.co, z.C)
            if self.raw:
                ops.nhwc_to_nchw(zs, out)
            else:
                ops.softmax_argmax(zs, out, None)

    def exchange(self):
        """slabs up to rank - 1, slabs of rank + 1 added on top of the own windows (in placement order)"""
        if self.world == 1 or self.plan.active == 1:
            return
        r, act = self.rank, self.plan.active
        if r >= act:
            return
        nrecv = self.plan.slab_floats(r + 1, self.C) if r + 1 < act else 0
        got = _exchange(self.sendbuf if r > 0 else None, r - 1, nrecv, r + 1, torch.float32, self.dev)
        if got is None:
            return
        off = 0
        for i, rr in self.plan.slabs(r + 1):
            y0, x0, _, w = (int(v) for v in self.plan.places[i])
            slab = got[off:off + self.C * rr * w].view(self.C, rr, w)
            off += self.C * rr * w
            if self.int8:
                self._add_int8(slab, y0 - self.lo, x0)
            else:
                ops.mosaic_accumulate(slab, self.mosaic, self.count, y0 - self.lo, x0)

    def finish(self, want):
        """want: "argmax" | "all" | int class index.  Returns (on rank 0) the full-size numpy array, None elsewhere."""
        rows, MW, MH = self.hi - self.lo, self.plan.MW, self.plan.MH
        if self.int8:
            merged, counter = self.mosaic[:, :rows].cpu().numpy(), self.count[:, :rows].cpu().numpy()
            m = counter > 0
            merged[m] //= counter[m]                        # predict.py:324-329: integer floor division, numpy semantics
            part = merged.argmax(axis=0).astype(np.uint8) if want == "argmax" else (merged if want == "all" else merged[want])
            part = torch.from_numpy(np.ascontiguousarray(part))
        else:
            am = torch.empty((max(rows, 1), MW), dtype=torch.uint8, device=self.dev) if want == "argmax" else None
            if rows > 0:
                ops.mosaic_finalize_rows(self.mosaic, self.count, 0, rows, am, fill=-9999.0 if self.raw else None)
            part = am[:rows] if want == "argmax" else (self.mosaic[:, :rows] if want == "all" else self.mosaic[want, :rows])
        return self._gather_rows(part, want == "all")

    def _gather_rows(self, part: torch.Tensor, planes: bool):
        """row strips of the ranks -> one array on rank 0 (only the requested band(s) travel)"""
        MW, MH = self.plan.MW, self.plan.MH
        if self.world == 1:
            return part.cpu().numpy()
        dist = _dist()
        host = dist.get_backend() == "gloo"
        part = (part.cpu() if host else part.to(self.dev)).contiguous()
        if self.rank != 0:
            if part.numel():
                dist.send(part, 0)
            return None
        shape = (self.C, MH, MW) if planes else (MH, MW)
        full = torch.empty(shape, dtype=part.dtype, device="cpu" if host else self.dev)
        sl = (slice(None),) if planes else ()
        full[sl + (slice(self.lo, self.hi),)] = part
        for r in range(1, self.plan.active):
            lo, hi = self.plan.own[r]
            if hi <= lo:
                continue
            buf = torch.empty(((self.C,) if planes else ()) + (hi - lo, MW), dtype=part.dtype, device=full.device)
            dist.recv(buf, r)
            full[sl + (slice(lo, hi),)] = buf
        return full.cpu().numpy()


def _run_merge(model, places, MH, MW, regression, int8_merge, rank, world, batch, make_input: Callable, want, timing: Optional[dict] = None):
    """make_input(first, n, n_pad) -> ops.WindowBatch of placements [first, first + n) padded to n_pad windows (so that every
    forward runs on ONE batch geometry and no second set of activation buffers is allocated)"""
    mg = _Merge(model, places, MH, MW, regression, int8_merge, rank, world, batch)
    t0 = time.perf_counter()
    done = 0
    batches = mg.plan.batches(rank, batch)
    n_pad = max((n for _, n in batches), default=0)
    for first, n in batches:
        wb = make_input(first, n, n_pad)
        z = model.forward_windows(wb)
        mg.add_batch(first, n, z)
        done += n
    mg.exchange()
    if timing is not None and not int8_merge and mg.hi > mg.lo:      # coverage of this rank's strip (before the division consumes nothing of it)
        timing.update(hits_min=int(mg.count.min().item()), hits_max=int(mg.count.max().item()))
    out = mg.finish(want)
    if timing is not None:
        torch.cuda.synchronize()
        timing.update(seconds=time.perf_counter() - t0, windows_this_rank=done, windows=len(mg.plan.places), active_ranks=mg.plan.active,
                      strip_rows=mg.hi - mg.lo, slab_floats_sent=0 if mg.sendbuf is None else mg.sendbuf.numel())
    return out


def _dist_ctx():
    from unet_amd.distributed import init_from_env
    return init_from_env()


def _want(regression, all_classes, specific_class):
    if regression:
        return 0                                      # the single band
    if all_classes:
        return "all"
    return "argmax" if specific_class is None else int(specific_class)


# ----------------------------------------------------------------------------------------------- configs[4]: a whole raster

def predict_raster(model, raster, size: int = 512, overlap: float = 0.2, *, max_empty: float = 0.9, dtype: str = "int8", nodata=None,
                   regression: bool = False, all_classes: bool = False, specific_class: Optional[int] = None, large_file: bool = False,
                   batch_size: int = 16, out_path=None, class_zero: bool = False, timing: Optional[dict] = None,
                   batch_invariant: bool = False):
    """Sliding-window prediction of a whole raster: equals split_raster(raster, patch_size=size, patch_overlap=overlap, max_empty) ->
    save_predictions(merge=True) on the tiles it writes (create_tiles_unet.py:252-434, predict.py:146-334).

    model    HipDynamicUnet (eval weights) or a Learner
    raster   path of a GeoTIFF, or an integer array [C, H, W] (numpy / torch, host or device)
    dtype    "int8" | "int16": the reference's DATATYPE switch -- int16 rasters are divided by 255 twice (utils.py:248-249 + IntToFloatTensor)
    batch_size  windows per forward launch.  Results are bit-reproducible for a FIXED batch_size (any rank count, any run); a different
             batch_size changes the launch grids and with them which deep-stage convs run as split reductions, i.e. the
             logits at rounding level (<= 2e-5 of the logit scale, tests/test_fullsize_gpu.py): masks can differ in numerical-tie pixels.
             The reference's own loop is batch 1 (predict.py:191-193)
    batch_invariant  True: every launch is planned as if its batch were ONE window (unet_tuning.plan_batch = 1), so each window runs exactly
             the kernels and split chains it would run alone: the result is bit-identical for every batch_size (and equals the
             tile-by-tile loop), at the price of batch-1 plans on full grids
    large_file  the reference's int8 merge (predict.py:209-214,288-289,324-329): probabilities as around(p * 31) in int8 rasters, int8 hit
             counters, integer floor division -- same numbers as save_predictions(merge=True, large_file=True)
    Returns on rank 0 the merged array (uint8 argmax [H', W'] by default; float32 [C, H', W'] for all_classes; one float32 plane for
    specific_class / regression; int8 planes with large_file) where H' x W' is the extent of the kept windows, None on the other ranks;
    with out_path it is also written as a GeoTIFF (class_zero shifts the class ids back, predict.py:19-52)."""
    model = getattr(model, "model", model)
    rank, local_rank, world = _dist_ctx()
    dev = model._device
    gt, tags = None, {}
    if isinstance(raster, (str, os.PathLike)):
        arr, meta = read_tiff(raster)
        gt, tags = meta["geotransform"], meta["tags"]
        nodata = meta.get("nodata") if nodata is None else nodata
        raster = arr[None] if arr.ndim == 2 else arr
    if isinstance(raster, np.ndarray):
        raster = _torch_samples(raster)
    if raster.dim() == 2:
        raster = raster[None]
    if raster.dtype not in ops.RASTER_TYPES:
        raster = raster.to(torch.int32)
    data = raster.to(dev).contiguous()
    if nodata is not None and data.data_ptr() == raster.data_ptr():
        data = data.clone()                                                        # never modify the caller's tensor
    src = ops.WindowSource(data, div255_twice=(dtype == "int16"))
    if nodata is not None:
        ops.raster_nodata_zero(src, nodata)                                        # create_tiles_unet.py:344-352
    Cb, H, W = src.C, src.H, src.W
    wins = sliding_windows(H, W, size, overlap)                                    # create_tiles_unet.py:52-54
    table_all = ops.window_table(wins.tolist(), dev)
    nz = ops.window_nonzero(src, table_all, size, size).cpu().numpy()
    keep = keep_windows(nz, Cb, size, size, max_empty)                             # create_tiles_unet.py:379
    wins = wins[keep]
    if len(wins) == 0:
        raise ValueError("every window of the raster is emptier than max_empty: nothing to predict")
    oy, ox = int(wins[:, 0].min()), int(wins[:, 1].min())                          # extent of the tiles present (predict.py:259-270)
    MH, MW = int(wins[:, 0].max()) + size - oy, int(wins[:, 1].max()) + size - ox
    places = np.concatenate([wins - np.array([oy, ox]), np.full((len(wins), 2), size, dtype=np.int64)], axis=1)
    # gather table in raster coordinates, padded at the end so that the last batch can be filled up with repeats of the last window
    rows = wins.tolist() + [wins[-1].tolist()] * batch_size
    gtab = ops.window_table(rows, dev)

    def make_input(first, n, n_pad):
        return ops.WindowBatch(src, gtab, first, n_pad, size, size)

    want = _want(regression, all_classes, specific_class)
    with (ops.tuning(plan_batch=1) if batch_invariant else contextlib.nullcontext()):
        out = _run_merge(model, places, MH, MW, regression, bool(large_file and not regression), rank, world, batch_size, make_input, want, timing)
    if rank == 0 and out_path is not None:
        ogt = None if gt is None else [gt[0] + ox * gt[1], gt[1], 0.0, gt[3] + oy * gt[5], 0.0, gt[5]]
        store_tif(out_path, out, ogt, tags, -9999 if regression else None, class_zero)
    if timing is not None:
        timing.update(kept_windows=len(wins), all_windows=int(len(keep)), mosaic=(MH, MW))
    return out if rank == 0 else None


# ----------------------------------------------------------------------------------------------- tile files: decode ahead of the GPU

class _TilePrefetcher:
    """A thread decodes the tile files of the coming batches into pinned host buffers; the main thread uploads the integer samples
    (asynchronous copy) and cuts / scales them on the device.  Yields (first, n, pinned tensor [n_pad, C, h, w]); the consumer hands a
    buffer back with the event recorded behind its upload, the producer waits for that event before it overwrites the buffer."""

    def __init__(self, tiles: Sequence[Path], batches: List[Tuple[int, int]], depth: int = 3):
        self.tiles, self.batches = tiles, batches
        self.n_pad = max((n for _, n in batches), default=0)
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.depth = depth
        self._free, self._made = {}, {}
        self.err = None
        self.t = threading.Thread(target=self._work, daemon=True)
        self.t.start()

    def _buf(self, key, shape, dt):
        fq = self._free.setdefault(key, queue.Queue())
        if fq.empty() and self._made.get(key, 0) < self.depth + 2:
            self._made[key] = self._made.get(key, 0) + 1
            return torch.empty(shape, dtype=dt, pin_memory=torch.cuda.is_available())
        buf, ev = fq.get()
        if ev is not None:
            ev.synchronize()
        return buf

    def _work(self):
        # tile files are decoded by a small pool (file read, strip copies and the band de-interleave release the GIL), `depth` batches ahead
        from concurrent.futures import ThreadPoolExecutor
        try:
            workers = max(2, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)))
        except OSError:
            workers = 4
        try:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                load = lambda path: np.ascontiguousarray(_as_samples(open_tile(path)))
                pending, nxt = [], 0
                for bi, (first, n) in enumerate(self.batches):
                    while nxt < len(self.batches) and nxt <= bi + self.depth:
                        f0, n0 = self.batches[nxt]
                        pending.append([ex.submit(load, self.tiles[f0 + j]) for j in range(n0)])
                        nxt += 1
                    arrs = [f.result() for f in pending.pop(0)]
                    self._emit(first, n, arrs)
            self.q.put(None)
        except BaseException as e:      # noqa: BLE001  (surfaces in the consumer)
            self.err = e
            self.q.put(None)

    def _emit(self, first, n, arrs):
        a0 = arrs[0]
        tdt = _torch_samples(a0[:0]).dtype
        buf = self._buf((a0.shape, tdt), (self.n_pad,) + a0.shape, tdt)
        for j, a in enumerate(arrs):
            buf[j].copy_(_torch_samples(a))
        self.q.put((first, n, buf))

    def __iter__(self):
        while True:
            it = self.q.get()
            if it is None:
                if self.err is not None:
                    raise self.err
                return
            yield it

    def upload(self, buf: torch.Tensor, device) -> torch.Tensor:
        """asynchronous host -> device copy of a yielded buffer; the buffer goes back to the producer behind the copy"""
        d = buf.to(device, non_blocking=True)
        ev = None
        if d.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._free[(tuple(buf.shape[1:]), buf.dtype)].put((buf, ev))
        return d


class _FeedPrefetcher:
    """The same interface on the training feed's machinery (unet_amd/feed.py): every pool task reads ONE tile file and copies it straight into
    its slot of a pinned staging buffer (the producer thread above copies the 16 tiles of a batch one after the other), uploads go out on a
    copy stream one batch ahead.  For tile sets of one height x width -- what split_raster writes (a tile with other bands or another sample
    type makes the feeder raise); sets of mixed sizes keep _TilePrefetcher.
    Yields (first, n, slot); `upload(slot, device)` returns the device buffer [n_pad, C, h, w] (rows beyond n: leftovers of an earlier batch,
    their windows are dropped); `done(slot)` after the forward that reads it has been issued."""

    def __init__(self, tiles: Sequence[Path], batches: List[Tuple[int, int]], device, depth: int = 3):
        from unet_amd.feed import BatchFeeder
        self.tiles, self.batches = tiles, batches
        self.n_pad = max((n for _, n in batches), default=0)
        self.feeder = BatchFeeder(lambda i: (_as_samples(open_tile(self.tiles[i])),), self.n_pad, device, depth=depth)

    def __iter__(self):
        for (first, n), slot in zip(self.batches, self.feeder.run([range(first, first + n) for first, n in self.batches])):
            yield first, n, slot

    @staticmethod
    def upload(slot, device) -> torch.Tensor:
        return slot.dev[0]

    @staticmethod
    def done(slot):
        slot.release()

    def close(self):
        self.feeder.close()


def _prefetcher(tiles, batches, sizes, device):
    """_FeedPrefetcher when every tile has the same height x width, else the general one"""
    if len(set(sizes)) == 1 and device.type == "cuda":
        return _FeedPrefetcher(tiles, batches, device)
    return _TilePrefetcher(tiles, batches)


def save_predictions(predict_model, predict_path, regression, merge=False, all_classes=False, specific_class=None, large_file=False,
                     AOI=None, year=None, validation_vision=True, class_zero=False, batch_size=16, timing: Optional[dict] = None,
                     batch_invariant: bool = False):
    rank, local_rank, world = _dist_ctx()
    dist = _dist()
    learn = load_learner(Path(predict_model), device=f"cuda:{local_rank}" if world > 1 else "cuda")
    model = learn.model
    path = Path(predict_path)
    output_folder = path.parent if merge else path.parent / ("predicted_tiles_" + Path(predict_model).stem)
    if rank == 0:
        output_folder.mkdir(parents=True, exist_ok=True)
    model_name = os.path.basename(predict_model).split(".")[0]
    tiles = sorted([p for p in path.iterdir() if p.suffix.lower() in (".tif", ".tiff", ".npy")])
    if rank == 0:
        print(f"Started at: {time.strftime('%H:%M:%S')}  ({len(tiles)} tiles)")
    dtype = learn.dls.train_ds.dtype
    geos = [_geo(t) for t in tiles]
    dev = model._device
    C = model.n_out
    div2 = dtype == "int16"
    t_start = time.perf_counter()

    if merge:
        # overlap merge (predict.py:257-355).  The extent follows from the tiles' geotransforms and sizes, which are known from the
        # headers BEFORE any tile is predicted: every batch is accumulated into the device mosaic as soon as it is computed and its
        # probabilities are dropped (the reference keeps all tiles' probabilities until the end).
        if any(g[0] is None for g in geos):
            raise ValueError("merge=True needs georeferenced tiles (.npy tiles carry no geotransform)")
        gts = np.array([[g[0][0], g[3], g[0][1], g[0][3], g[2], g[0][5]] for g in geos], dtype=np.float64)
        ulx_full, uly_full = gts[:, 0].min(), gts[:, 3].max()
        xres, yres = gts[0, 2], gts[0, 5]
        xmax_i, ymin_i = gts[:, 0].argmax(), gts[:, 3].argmin()
        lrx_full = gts[:, 0].max() + gts[xmax_i, 1] * gts[xmax_i, 2]
        lry_full = gts[:, 3].min() + gts[ymin_i, 4] * gts[ymin_i, 5]
        if len(set(gts[:, 1])) != 1 or len(set(gts[:, 4])) != 1:
            warnings.warn("Not all tiles have the same resolution.")
        MW, MH = round((lrx_full - ulx_full) / xres), round((lry_full - uly_full) / yres)
        int8_merge = bool(large_file and not regression)
        if rank == 0:
            print(f"True merged raster size: {C * MH * MW * (1 if int8_merge else 4) / (1024 ** 2): .1f}MB.")
        places = np.array([[round((gts[i, 3] - uly_full) / yres), round((gts[i, 0] - ulx_full) / xres), geos[i][2], geos[i][3]]
                           for i in range(len(tiles))], dtype=np.int64)
        order = merge_order(places)
        places, tiles_o = places[order], [tiles[i] for i in order]
        plan = MergePlan(places, MH, MW, world)
        pf = _prefetcher(tiles_o, plan.batches(rank, batch_size), [(int(p_[2]), int(p_[3])) for p_ in places], dev)
        feed = iter(pf)
        ztab = ops.window_table([[0, 0, j, 0] for j in range(batch_size)], dev)
        held = []          # staging slot of the batch in flight: released once the NEXT batch is asked for (its gather has been issued by then)

        def make_input(first, n, n_pad):
            while held:
                pf.done(held.pop())
            f, nn, buf = next(feed)
            d = pf.upload(buf, dev)
            assert (f, nn) == (first, n) and d.shape[0] == n_pad, ((f, nn, d.shape[0]), (first, n, n_pad))
            if hasattr(pf, "done"):
                held.append(buf)
            return ops.WindowBatch(ops.WindowSource(d, div255_twice=div2), ztab, 0, n_pad, d.shape[2], d.shape[3])

        want = _want(regression, all_classes, specific_class)
        try:
            with (ops.tuning(plan_batch=1) if batch_invariant else contextlib.nullcontext()):          # (see predict_raster)
                out = _run_merge(model, places, MH, MW, regression, int8_merge, rank, world, batch_size, make_input, want, timing)
        finally:
            feed.close()
            if hasattr(pf, "close"):
                pf.close()              # (the decode pool and its pinned ring live for one call)
        if timing is not None:
            timing["tiles_per_s_end_to_end"] = len(tiles) / (time.perf_counter() - t_start)
        if rank != 0:
            return None
        name = "_".join(filter(None, [AOI, year, model_name, "prediction"])) + ".tif"
        store_tif(output_folder / name, out, [ulx_full, xres, 0.0, uly_full, 0.0, yres], geos[0][1], -9999 if regression else None,
                  class_zero)
        print(f"Prediction stored in {output_folder}.")
        return output_folder if regression else output_folder / name

    # ---- one output file per tile (predict.py:224-254); tile i -> rank i mod world
    mine = list(range(rank, len(tiles), world))
    sizes = [(geos[i][2], geos[i][3]) for i in mine]
    batches, i = [], 0
    while i < len(mine):
        n = 1
        while n < batch_size and i + n < len(mine) and sizes[i + n] == sizes[i]:
            n += 1
        batches.append((i, n))
        i += n
    mtiles = [tiles[i] for i in mine]
    pf = _prefetcher(mtiles, batches, sizes, dev)
    ztab = ops.window_table([[0, 0, j, 0] for j in range(batch_size)], dev)
    for first, n, buf in pf:
        d = pf.upload(buf, dev)
        wb = ops.WindowBatch(ops.WindowSource(d, div255_twice=div2), ztab, 0, d.shape[0], d.shape[2], d.shape[3])
        with (ops.tuning(plan_batch=1) if batch_invariant else contextlib.nullcontext()):
            z = model.forward_windows(wb)
        if hasattr(pf, "done"):
            pf.done(buf)               # (the gather that reads the staging buffer has been issued)
        zs = ops.TS(z.buf[:n], z.co, z.C)
        if regression:       # predict.py:195-197: tile_preds[1] = raw outputs [1,H,W]
            probs, amax = torch.empty((n, C, z.H, z.W), dtype=torch.float32, device=dev), None
            ops.nhwc_to_nchw(zs, probs)
        else:
            need_p = all_classes or specific_class is not None
            probs = torch.empty((n, C, z.H, z.W), dtype=torch.float32, device=dev) if need_p else None
            amax = None if need_p else torch.empty((n, z.H, z.W), dtype=torch.int64, device=dev)
            ops.softmax_argmax(zs, probs, amax)
        outs = (probs if probs is not None else amax.to(torch.uint8)).cpu().numpy()
        for j in range(n):
            t = mtiles[first + j]
            gt, tags = geos[mine[first + j]][0], geos[mine[first + j]][1]
            if regression or all_classes:
                out = outs[j]
            elif specific_class is None:
                out = outs[j]
            else:
                out = outs[j, specific_class]
            if large_file and out.dtype.kind == "f" and out.max() <= 1 and (all_classes or specific_class):
                out = np.around(out * LARGE_FILE_SCALE).astype(np.int8)
            name = t.name if t.suffix != ".npy" else t.stem + ".tif"
            store_tif(output_folder / name, out, gt, tags, None, class_zero)
    if hasattr(pf, "close"):
        pf.close()
    if validation_vision:
        pass  # per-tile majority-class confusion plots (predict.py:56-143) are reporting, out of scope
    if world > 1:
        dist.barrier()
    if timing is not None:
        timing["tiles_per_s_end_to_end"] = len(tiles) / (time.perf_counter() - t_start)
    if rank == 0:
        print(f"Prediction stored in {output_folder}.")
    return output_folder
