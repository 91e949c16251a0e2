"""CPU suite: the row-block merge plan of predict_raster / save_predictions(merge=True) (unet_amd/mosaic.py).

The device kernels are not called here; the N-rank schedule -- own windows in order, then the neighbour's slabs in order, strips
finalised per rank -- is emulated in numpy float32 and must equal the single-process merge (predict.py:284-334) bit for bit."""
import numpy as np
import pytest

from unet_amd.mosaic import MergePlan, keep_windows, merge_order, sliding_windows, window_offsets


def test_window_rule_cfg5():
    w = sliding_windows(20000, 20000, 512, 0.2)
    ys = sorted(set(w[:, 0].tolist()))
    assert len(w) == 49 * 49 and ys[:3] == [0, 410, 820] and ys[-2:] == [19270, 19488]
    assert w[:3].tolist() == [[0, 0], [0, 410], [0, 820]]                      # row-major: the reference's tile index order
    assert window_offsets(1000, 400, 400) == [0, 400, 600] and window_offsets(800, 400, 400) == [0, 400]
    with pytest.raises(ValueError):
        sliding_windows(300, 800, 400, 0.2)


def test_emptiness_filter_is_the_reference_inequality():
    # create_tiles_unet.py:379: skipped when sum(crop != 0) < prod(crop.shape) * (1 - max_empty)
    nz = np.array([0, 99, 100, 101, 1000])
    assert keep_windows(nz, 4, 5, 50, 0.9).tolist() == [False, False, True, True, True]
    thr = np.prod((5, 50, 4)) * (1 - 0.9)
    assert [(int(v) < thr) for v in nz] == [True, True, thr > 100, False, False]


def _single(places, probs, MH, MW, C):
    m, c = np.zeros((C, MH, MW), np.float32), np.zeros((MH, MW), np.int32)
    for (y, x, h, w), p in zip(places, probs):
        m[:, y:y + h, x:x + w] += p
        c[y:y + h, x:x + w] += 1
    return m, c


def _emulate(plan: MergePlan, probs, C):
    """what the ranks do, in numpy: returns the mosaic assembled from the strips"""
    MH, MW = plan.MH, plan.MW
    full_m, full_c = np.zeros((C, MH, MW), np.float32), np.zeros((MH, MW), np.int32)
    strips = []
    for r in range(plan.world):
        lo, hi = plan.own[r]
        m, c = np.zeros((C, hi - lo, MW), np.float32), np.zeros((hi - lo, MW), np.int32)
        a, b = plan.ranges[r]
        for i in range(a, b):
            y, x, h, w = plan.places[i]
            r0, r1 = max(y, lo), min(y + h, hi)
            if r1 > r0:
                m[:, r0 - lo:r1 - lo, x:x + w] += probs[i][:, r0 - y:r1 - y]
                c[r0 - lo:r1 - lo, x:x + w] += 1
        strips.append((m, c))
    for r in range(plan.world - 1):            # slabs of rank r + 1 land on rank r AFTER its own windows
        lo, hi = plan.own[r]
        m, c = strips[r]
        for i, rows in plan.slabs(r + 1):
            y, x, h, w = plan.places[i]
            assert lo <= y and y + rows <= hi, "a slab must lie inside the strip of the rank before"
            m[:, y - lo:y - lo + rows, x:x + w] += probs[i][:, :rows]
            c[y - lo:y - lo + rows, x:x + w] += 1
    for r in range(plan.world):
        lo, hi = plan.own[r]
        full_m[:, lo:hi], full_c[lo:hi] = strips[r]
    return full_m, full_c


@pytest.mark.parametrize("H,W,size,overlap,world", [(1500, 1300, 512, 0.2, 2), (1500, 1300, 512, 0.2, 3), (2000, 700, 256, 0.5, 4),
                                                      (900, 900, 400, 0.2, 8), (3000, 600, 200, 0.6, 5)])
def test_row_block_schedule_equals_single_process_bit_for_bit(H, W, size, overlap, world):
    g = np.random.default_rng(H + world)
    wins = sliding_windows(H, W, size, overlap)
    places = np.concatenate([wins, np.full((len(wins), 2), size)], axis=1)
    C = 3
    probs = [g.random((C, size, size), dtype=np.float32) for _ in places]
    ref_m, ref_c = _single(places, probs, H, W, C)
    plan = MergePlan(places, H, W, world)
    assert plan.ranges[0][0] == 0 and plan.ranges[-1][1] == len(places)
    assert all(a[1] == b[0] for a, b in zip(plan.ranges[:-1], plan.ranges[1:])) and all(a[1] == b[0] for a, b in zip(plan.own[:-1], plan.own[1:]))
    assert plan.own[0][0] == 0 and plan.own[plan.active - 1][1] == H
    got_m, got_c = _emulate(plan, probs, C)
    assert np.array_equal(got_c, ref_c) and np.array_equal(got_m, ref_m)       # order of the float32 additions is preserved
    sizes = [b - a for a, b in plan.ranges[:plan.active]]
    assert max(sizes) - min(sizes) <= 1                                         # balanced by tile count


def test_ranks_are_dropped_when_windows_would_reach_two_strips_up():
    # 3 window rows, 8 ranks: equal cuts would put one window row on several ranks each; the plan falls back to fewer active ranks
    wins = sliding_windows(1100, 512, 512, 0.5)
    places = np.concatenate([wins, np.full((len(wins), 2), 512)], axis=1)
    plan = MergePlan(places, 1100, 512, 8)
    assert 1 <= plan.active < 8 and plan.ranges[plan.active] == (len(places), len(places))
    g = np.random.default_rng(1)
    probs = [g.random((2, 512, 512), dtype=np.float32) for _ in places]
    m, c = _emulate(plan, probs, 2)
    rm, rc = _single(places, probs, 1100, 512, 2)
    assert np.array_equal(m, rm) and np.array_equal(c, rc)


def test_cfg5_plan_traffic():
    wins = sliding_windows(20000, 20000, 512, 0.2)
    places = np.concatenate([wins, np.full((len(wins), 2), 512)], axis=1)
    plan = MergePlan(places, 20000, 20000, 8)
    assert plan.active == 8 and [b - a for a, b in plan.ranges] == [300, 300, 300, 300, 300, 300, 300, 301]
    worst = max(plan.slab_floats(r, 5) for r in range(8)) * 4
    assert worst < 320e6                    # per boundary: < 320 MB of slabs, against 8 GB + 1.6 GB for an all-reduce of whole mosaics
    assert sum(hi - lo for lo, hi in plan.own) == 20000


def test_arbitrary_tile_sets_are_sorted_and_batched_by_size():
    places = np.array([[100, 0, 64, 64], [0, 64, 64, 64], [0, 0, 64, 64], [100, 64, 32, 32], [40, 10, 64, 64]])
    o = merge_order(places)
    assert places[o][:, :2].tolist() == [[0, 0], [0, 64], [40, 10], [100, 0], [100, 64]]
    plan = MergePlan(places[o], 164, 128, 1)
    assert plan.batches(0, 16) == [(0, 4), (4, 1)]                              # the 32 x 32 tile cannot share a launch with 64 x 64 ones
    assert plan.batches(0, 3) == [(0, 3), (3, 1), (4, 1)]


# ------------------------------------------------------------------------------------------------ the exchange itself, two gloo ranks on the CPU

def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _exchange_worker(rank, world, port, q, geom=(1500, 700, 512)):
    """what the ranks of predict._Merge do between the forward passes and the finalisation, with the device kernels replaced by numpy:
    slabs travel through predict._exchange (isend / irecv pair), strips through predict._Merge._gather_rows (send / recv to rank 0)"""
    import os
    import types
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import predict as P
    (H, W, size), C = geom, 3
    wins = sliding_windows(H, W, size, 0.2)
    places = np.concatenate([wins, np.full((len(wins), 2), size)], axis=1)
    g = np.random.default_rng(5)
    probs = [g.random((C, size, size), dtype=np.float32) for _ in places]           # the same "predictions" on both ranks
    plan = MergePlan(places, H, W, world)
    assert plan.active == world, (plan.active, world)          # the geometry is chosen so that every rank owns a strip
    lo, hi = plan.own[rank]
    m, c = np.zeros((C, hi - lo, W), np.float32), np.zeros((hi - lo, W), np.int32)
    a, b = plan.ranges[rank]
    for i in range(a, b):
        y, x, h, w = plan.places[i]
        r0, r1 = max(y, lo), min(y + h, hi)
        if r1 > r0:
            m[:, r0 - lo:r1 - lo, x:x + w] += probs[i][:, r0 - y:r1 - y]
            c[r0 - lo:r1 - lo, x:x + w] += 1
    send = None
    if plan.slabs(rank):
        send = torch.from_numpy(np.concatenate([probs[i][:, :rows].ravel() for i, rows in plan.slabs(rank)]))
    nrecv = plan.slab_floats(rank + 1, C) if rank + 1 < plan.active else 0
    got = P._exchange(send if rank > 0 else None, rank - 1, nrecv, rank + 1, torch.float32, "cpu")
    if got is not None:
        off = 0
        for i, rows in plan.slabs(rank + 1):
            y, x, h, w = plan.places[i]
            slab = got[off:off + C * rows * w].view(C, rows, w).numpy()
            off += C * rows * w
            m[:, y - lo:y - lo + rows, x:x + w] += slab
            c[y - lo:y - lo + rows, x:x + w] += 1
    m[:, c > 0] /= c[c > 0]
    me = types.SimpleNamespace(plan=plan, rank=rank, world=world, lo=lo, hi=hi, C=C, dev="cpu")
    full = P._Merge._gather_rows(me, torch.from_numpy(m.argmax(0).astype(np.uint8)), False)
    allc = P._Merge._gather_rows(me, torch.from_numpy(m), True)
    if rank == 0:
        rm = np.zeros((C, H, W), np.float32); rc = np.zeros((H, W), np.int32)
        for (y, x, h, w), p in zip(places, probs):
            rm[:, y:y + h, x:x + w] += p
            rc[y:y + h, x:x + w] += 1
        rm[:, rc > 0] /= rc[rc > 0]
        q.put((rank, bool(np.array_equal(full, rm.argmax(0).astype(np.uint8))), bool(np.array_equal(allc, rm))))
    else:
        q.put((rank, full is None, allc is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,geom", [(2, (1500, 700, 512)), (8, (1400, 300, 128))])
def test_slab_exchange_and_strip_gather_gloo_ranks(world, geom):
    """world 8 = the rank count of BASELINE configs[4] (8-GPU tile-sharded predict): seven simultaneous isend / irecv boundaries, every
    rank receiving from the rank below while sending to the rank above, then the strip gather on rank 0"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q, geom)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert res == [(r, True, True) for r in range(world)], res
