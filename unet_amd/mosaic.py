"""Overlap merge of per-window predictions into one mosaic, partitioned by ROWS over the ranks.

Reference (single process, host): ``predict.py:257-334`` -- a ``[C, H, W]`` float mosaic of summed softmax probabilities plus a hit
counter, every tile added at ``round((ulx - ulx_full) / xres)``, divided where the counter is positive, argmax over the classes.  The
windows themselves follow ``slidingwindow.generate`` as ``create_tiles_unet.py:30-56`` calls it.

Here the merge order is fixed -- placements sorted by (row, column) -- so that every mosaic pixel receives its contributions in one
defined order whatever the number of ranks:

* the ordered placement list is cut into ``world`` contiguous ranges of equal length (tiles are the unit of work);
* rank r OWNS the mosaic rows from the end of rank r-1's coverage to the end of its own coverage, and keeps only that strip in HBM;
* a placement of rank r that starts above its strip (it overlaps rows owned by rank r-1) contributes those rows as a *slab* of per-window
  probabilities that is sent to rank r-1, which adds the slabs in order AFTER its own windows -- exactly the order a single process uses,
  so the N-rank result equals the 1-rank result bit for bit, and the only exchange is the overlap rows (cfg5: <= 300 MB per boundary
  instead of an 8 GB all-reduce of whole mosaics);
* every rank finalises its strip (divide, argmax) and only the requested band(s) travel to rank 0.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def window_offsets(length: int, size: int, step: int) -> List[int]:
    """slidingwindow 0.0.14 as create_tiles_unet.py:52-54 calls it: 0, step, ... while the window fits, plus one window flush with the
    far edge when the raster is not covered (20000 px, 512 px windows, overlap 0.2 -> 0, 410, ..., 19270, 19488: 49 per axis)"""
    offs = list(range(0, length - size + 1, step))
    if offs[-1] + size < length:
        offs.append(length - size)
    return offs


def sliding_windows(height: int, width: int, size: int, overlap: float) -> np.ndarray:
    """int64 [n, 2] = (y0, x0) of every window, row-major (the index order of the reference's tile files)"""
    if overlap > 1:
        raise ValueError(f"Patch overlap {overlap} must be between 0 - 1")
    if height < size or width < size:
        raise ValueError(f"Patch size of {size} is larger than the image dimensions {[height, width]}")
    step = size - int(size * overlap)
    ys, xs = window_offsets(height, size, step), window_offsets(width, size, step)
    return np.array([(y, x) for y in ys for x in xs], dtype=np.int64).reshape(-1, 2)


def keep_windows(nonzero: np.ndarray, bands: int, size_h: int, size_w: int, max_empty: float) -> np.ndarray:
    """create_tiles_unet.py:379: a window is dropped when np.sum(crop != 0) < np.prod(crop.shape) * (1 - max_empty)"""
    return ~(np.asarray(nonzero, dtype=np.int64) < np.prod((size_h, size_w, bands)) * (1 - max_empty))


def merge_order(places: np.ndarray) -> np.ndarray:
    """permutation that sorts placements [n, >=2] = (y0, x0, ...) by row, then column (stable)"""
    p = np.asarray(places)
    return np.lexsort((p[:, 1], p[:, 0]))


class MergePlan:
    """places: int [n, 4] = (y0, x0, h, w) in mosaic pixel coordinates, already in merge order (sorted by y0, then x0)."""

    def __init__(self, places: np.ndarray, MH: int, MW: int, world: int = 1):
        p = np.asarray(places, dtype=np.int64).reshape(-1, 4)
        if len(p) and (np.diff(p[:, 0]) < 0).any():
            raise ValueError("placements must be sorted by row (merge_order)")
        self.places, self.MH, self.MW, self.world = p, int(MH), int(MW), int(world)
        n = len(p)
        end = p[:, 0] + p[:, 2]
        for active in range(max(1, min(world, n)), 0, -1):
            cuts = [n * r // active for r in range(active + 1)]
            own_lo, own_hi = [0] * active, [0] * active
            for r in range(active):
                own_lo[r] = 0 if r == 0 else own_hi[r - 1]
                own_hi[r] = self.MH if r == active - 1 else max(own_lo[r], int(end[cuts[r]:cuts[r + 1]].max()))
            # a rank's windows may reach into the strip of the rank before it, never further up
            if all(int(p[cuts[r], 0]) >= own_lo[r - 1] for r in range(1, active)):
                break
        self.active = active
        self.ranges: List[Tuple[int, int]] = [(cuts[r], cuts[r + 1]) for r in range(active)] + [(n, n)] * (world - active)
        self.own: List[Tuple[int, int]] = [(own_lo[r], own_hi[r]) for r in range(active)] + [(self.MH, self.MH)] * (world - active)

    def slabs(self, rank: int) -> List[Tuple[int, int]]:
        """[(placement index, rows)] of rank's placements that start above its strip: their first `rows` rows belong to rank - 1"""
        if rank == 0 or rank >= self.active:
            return []
        a, b = self.ranges[rank]
        lo = self.own[rank][0]
        out = []
        for i in range(a, b):
            y0, _, h, _ = self.places[i]
            if y0 >= lo:
                break           # sorted by y0: nothing further starts above the strip
            out.append((i, int(min(y0 + h, lo) - y0)))
        return out

    def slab_floats(self, rank: int, C: int) -> int:
        return int(sum(C * rows * int(self.places[i, 3]) for i, rows in self.slabs(rank)))

    def batches(self, rank: int, batch: int) -> List[Tuple[int, int]]:
        """(first, n) runs of at most `batch` consecutive placements of one size inside the rank's range"""
        a, b = self.ranges[rank]
        out, i = [], a
        while i < b:
            n = 1
            while n < batch and i + n < b and tuple(self.places[i + n, 2:]) == tuple(self.places[i, 2:]):
                n += 1
            out.append((i, n))
            i += n
        return out
