"""Tile-batch data parallelism: one process per GPU, every rank trains on its own tiles, gradients are
summed with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference has no distributed code; this is the build-defined N>1 path of
BASELINE.json configs[2].  Design for xGMI (point-to-point links, per-link bound rings): the payload is the
single flat fp32 gradient buffer (165 MB for xresnet34), reduced in a few LARGE buckets that follow the
backward order -- the decoder span is complete when the decoder backward ends and is reduced on RCCL's
stream while the encoder backward still runs; the encoder span follows.  Gradients are pre-scaled by
1/world in the loss kernel, so the collective is a plain SUM.
BatchNorm statistics stay per replica (the reference has no SyncBN).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # UNET_DIST_BACKEND=gloo lets several ranks share ONE GPU for rehearsals (RCCL refuses duplicate devices)
            backend = os.environ.get("UNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("UNET_FORCE_DEVICE") is not None:
            local_rank = int(os.environ["UNET_FORCE_DEVICE"])
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def bucket_spans(total: int, boundaries: List[int], max_bucket: int) -> List[Tuple[int, int]]:
    """Split [0,total) at `boundaries` (readiness points of the backward) and further into <= max_bucket pieces.
    Returned in REVERSE (backward) order: last span first."""
    cuts = sorted(set([0, total] + [b for b in boundaries if 0 < b < total]))
    spans: List[Tuple[int, int]] = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        n = b - a
        k = max(1, -(-n // max_bucket))
        step = -(-n // k)
        s = a
        while s < b:
            e = min(b, s + step)
            spans.append((s, e))
            s = e
    return spans[::-1]


class GradReducer:
    """Bucketed asynchronous SUM all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, boundaries: List[int], max_bucket_elems: int = 16 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.spans = bucket_spans(flat_grad.numel(), boundaries, max_bucket_elems)
        self._pending = []
        self._next = 0

    def reset(self):
        self._pending = []
        self._next = 0

    def ready_down_to(self, offset: int):
        """every gradient element at index >= offset is final: launch the buckets that lie fully above it"""
        if self.world == 1:
            return
        while self._next < len(self.spans) and self.spans[self._next][0] >= offset:
            a, b = self.spans[self._next]
            self._pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._next += 1

    def finish(self):
        """launch whatever is left and make the current stream wait for all buckets"""
        self.ready_down_to(0)
        for w in self._pending:
            w.wait()
        self.reset()


def broadcast_parameters(flat_param: torch.Tensor, buffers, src: int = 0, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat_param, src, group=group)
    for b in buffers:
        dist.broadcast(b, src, group=group)
