"""Batch feed of the training / validation loop: decode pool -> pinned INTEGER staging -> asynchronous upload ahead of the GPU.

The reference's loader (``train.py:345`` -> fastai ``DataLoader(num_workers=0)`` -> ``data.py:18-28`` ``open_npy`` per item ->
``utils.py:239-295`` batch transform) reads, scales and stacks every batch on the thread that launches the step.  Here the step is
24-110 ms of GPU work, so the feed must never be on that thread's critical path:

* a pool of worker threads reads the tile files of the coming ``depth`` batches (file reads, strip decoders and the band de-interleave
  all release the GIL) straight into slot ``j`` of a pinned staging buffer, as the samples the file holds (uint8 / uint16 / ...: one or
  two bytes per sample instead of four, masks one byte instead of eight);
* the launch thread issues ONE ``hipMemcpyAsync`` per tensor on a copy stream, one batch ahead when the decoders keep up;
* value scaling (``/255`` [``/255``]), the int64 widening of the mask and the flips happen on the device (``unet_tiles_stage`` /
  ``unet_mask_stage``, csrc/raster.hip) -- bit-equal to the host arithmetic of ``learner.scale_input``.

Buffers are recycled behind events: a pinned buffer is refilled only after its upload has completed, a device staging buffer is
overwritten only after the kernels that read it have run.
"""
from __future__ import annotations

import os
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

# numpy sample types the device kernels read as they are (ops.RASTER_TYPES); anything else is staged as int32, which is what data.py:24
# does to every tile anyway
_SAMPLE_TYPES = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16, np.dtype(np.int16): torch.int16,
                 np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}


def as_samples(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype.byteorder == ">" or (a.dtype.byteorder == "=" and not np.little_endian):
        a = a.astype(a.dtype.newbyteorder("<"))
    return a if a.dtype in _SAMPLE_TYPES else a.astype(np.int32)


def default_workers() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 2
    return max(2, min(16, n))


class _Slot:
    """pinned host buffers of one batch and their device twins"""

    def __init__(self, specs: Sequence[Tuple[tuple, np.dtype]], bs: int, device: torch.device):
        self.host, self.host_np, self.dev = [], [], []
        cuda = device.type == "cuda"
        for shape, dt in specs:
            nbytes = int(bs * int(np.prod(shape)) * dt.itemsize)
            h = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=cuda)
            self.host.append(h)
            self.host_np.append(h.numpy()[:nbytes].view(dt).reshape((bs,) + tuple(shape)))
            d = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device) if cuda else h
            self.dev.append(d[:nbytes].view(_SAMPLE_TYPES[dt]).view((bs,) + tuple(shape)))
        self.h2d_done: Optional[torch.cuda.Event] = None     # the upload of the batch in this slot
        self.consumed: Optional[torch.cuda.Event] = None     # the kernels that read the device twins
        self.futures: list = []
        self.n = 0

    def release(self):
        """called by the consumer after it has ISSUED its last kernel on the device twins (current stream)"""
        if self.dev and self.dev[0].is_cuda:
            self.consumed = torch.cuda.Event()
            self.consumed.record()


class BatchFeeder:
    """``load(i) -> tuple of arrays`` (e.g. image samples [C,H,W], mask [H,W]); ``run(batches)`` yields a ``_Slot`` per batch whose
    ``dev[k][:slot.n]`` hold the batch on ``device``.  One feeder per loader; threads and pinned buffers persist across epochs."""

    def __init__(self, load: Callable[[int], tuple], bs: int, device, depth: int = 3, workers: Optional[int] = None):
        self.load, self.bs, self.device, self.depth = load, int(bs), torch.device(device), max(1, int(depth))
        self.workers = workers or default_workers()
        self._ex: Optional[ThreadPoolExecutor] = None
        self._slots: List[_Slot] = []
        self._specs = None
        self._copy = None

    # -- lazily built, kept for the life of the loader
    def _pool(self) -> ThreadPoolExecutor:
        if self._ex is None:
            self._ex = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="unet-feed")
        return self._ex

    def _ring(self, probe_index: int) -> List[_Slot]:
        if not self._slots:
            arrs = [as_samples(a) for a in self.load(probe_index)]
            self._specs = [(tuple(a.shape), a.dtype) for a in arrs]
            self._slots = [_Slot(self._specs, self.bs, self.device) for _ in range(self.depth + 2)]
            if self.device.type == "cuda":
                self._copy = torch.cuda.Stream(self.device)
        return self._slots

    def _fill(self, slot: _Slot, j: int, index: int):
        arrs = self.load(index)
        if len(arrs) != len(self._specs):
            raise ValueError(f"item {index}: {len(arrs)} arrays, {len(self._specs)} expected")
        for k, a in enumerate(arrs):
            a = as_samples(a)
            shape, dt = self._specs[k]
            if tuple(a.shape) != shape or a.dtype != dt:
                raise ValueError(f"item {index}: tile of shape {tuple(a.shape)} / {a.dtype} in a dataset of {shape} / {dt} tiles "
                                 "(a batch is one tensor: every tile of a dataset must have the same size and sample type)")
            np.copyto(slot.host_np[k][j], a)

    def _submit(self, slot: _Slot, items: Sequence[int]):
        if slot.h2d_done is not None:           # the previous upload out of these pinned buffers (long complete: depth + 2 batches ago)
            slot.h2d_done.synchronize()
            slot.h2d_done = None
        slot.n = len(items)
        ex = self._pool()
        slot.futures = [ex.submit(self._fill, slot, j, int(i)) for j, i in enumerate(items)]

    def _upload(self, slot: _Slot):
        for f in slot.futures:
            f.result()                          # decode errors surface here, on the consumer's thread
        slot.futures = []
        if self._copy is None:
            return
        if slot.consumed is not None:
            self._copy.wait_event(slot.consumed)
            slot.consumed = None
        with torch.cuda.stream(self._copy):
            for h, d in zip(slot.host, slot.dev):
                nb = slot.n * (d[0].numel() * d.element_size())
                d.view(-1).view(torch.uint8)[:nb].copy_(h[:nb], non_blocking=True)
            slot.h2d_done = torch.cuda.Event()
            slot.h2d_done.record(self._copy)

    def run(self, batches: Sequence[Sequence[int]]):
        nb = len(batches)
        if nb == 0:
            return
        ring = self._ring(int(batches[0][0]))
        R = len(ring)
        submitted = uploaded = 0          # batches [0, submitted) are with the pool, [0, uploaded) are on their way to the device
        # UNET_FEED_SWITCH_S (opt-in): CPython's GIL switch interval while the feeder runs.  The decode threads hold the GIL only between their
        # C calls, but each time they ask for it back they wait up to the interval (5 ms by default) while the consumer thread issues launches.
        # Measured (scripts/ab_feed_switch.py, profiles/r05_ab_feed_switch.log; bf16 storage, fp32 does not move): 0.5 ms lifts JPEG tiles
        # 582 -> 614 tiles/s and LZW 604 -> 617, and costs uncompressed tiles 642 -> 600 (the launch thread is interrupted more often) --
        # uncompressed is what the reference's tiler writes, so the interpreter's setting is left alone unless asked.
        old_switch = sys.getswitchinterval()
        want = float(os.environ.get("UNET_FEED_SWITCH_S", "0") or 0)
        if 0.0 < want < old_switch:
            sys.setswitchinterval(want)
        try:
            for k in range(nb):
                while submitted < nb and submitted <= k + self.depth:
                    self._submit(ring[submitted % R], batches[submitted])
                    submitted += 1
                if uploaded <= k:
                    self._upload(ring[k % R])
                    uploaded = k + 1
                if uploaded == k + 1 and k + 1 < submitted and all(f.done() and f.exception() is None for f in ring[(k + 1) % R].futures):
                    self._upload(ring[(k + 1) % R])          # one batch ahead whenever the decoders keep up
                    uploaded = k + 2
                slot = ring[k % R]
                if slot.h2d_done is not None:
                    torch.cuda.current_stream().wait_event(slot.h2d_done)
                yield slot
        finally:
            sys.setswitchinterval(old_switch)
            for s in ring:                 # a consumer that stops early: nothing may write into the slots behind its back
                for f in s.futures:
                    f.cancel()
                for f in s.futures:
                    if not f.cancelled():
                        try:
                            f.result()
                        except Exception:      # noqa: BLE001
                            pass
                s.futures = []

    def close(self):
        if self._ex is not None:
            self._ex.shutdown(wait=True)
            self._ex = None
