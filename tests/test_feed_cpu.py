"""Host side of the training feed (unet_amd/feed.py) without a GPU: ordering, ragged batches, recycling of the staging ring over many
batches, error propagation from the decode pool, early exit; the flip draws the device feed hands to its kernels."""
import numpy as np
import pytest
import torch

from unet_amd.feed import BatchFeeder, as_samples


def _load_factory(n, fail_at=None):
    def load(i):
        if fail_at is not None and i == fail_at:
            raise OSError(f"tile {i} is unreadable")
        return np.full((2, 4, 6), i % 251, np.uint8), np.full((4, 6), (i * 3) % 7, np.uint8)
    return load


def test_batches_arrive_in_order_with_the_ring_recycled():
    f = BatchFeeder(_load_factory(200), bs=4, device="cpu", depth=2, workers=5)
    batches = [list(range(b * 4, min(b * 4 + 4, 198))) for b in range(50)]
    seen = 0
    for k, slot in enumerate(f.run(batches)):
        assert slot.n == len(batches[k])
        img, mask = slot.dev[0][:slot.n], slot.dev[1][:slot.n]
        assert img.dtype == torch.uint8 and tuple(img.shape) == (slot.n, 2, 4, 6)
        for j, i in enumerate(batches[k]):
            assert int(img[j, 0, 0, 0]) == i % 251 and int(mask[j, 0, 0]) == (i * 3) % 7
        slot.release()
        seen += slot.n
    assert seen == 198
    assert len(f._slots) == 4          # depth + 2 staging sets, however many batches pass through
    assert list(f.run([])) == []
    f.close()


def test_decode_errors_surface_in_the_consumer_and_early_exit_is_clean():
    f = BatchFeeder(_load_factory(40, fail_at=13), bs=4, device="cpu", depth=3, workers=3)
    batches = [list(range(b * 4, b * 4 + 4)) for b in range(10)]
    got = 0
    with pytest.raises(OSError, match="tile 13"):
        for slot in f.run(batches):
            got += 1
    assert got == 3
    # a consumer that stops after one batch: the generator's cleanup cancels / drains what the pool still holds
    f2 = BatchFeeder(_load_factory(40), bs=4, device="cpu", depth=3, workers=3)
    for slot in f2.run(batches):
        break
    assert all(not s.futures for s in f2._slots)
    with pytest.raises(ValueError, match="same size"):
        bad = BatchFeeder(lambda i: (np.zeros((2, 4 + (i == 5), 6), np.uint8),), bs=2, device="cpu")
        list(bad.run([[0, 1], [4, 5]]))


def test_sample_types_pass_or_go_through_int32():
    assert as_samples(np.zeros(3, np.uint16)).dtype == np.uint16
    assert as_samples(np.zeros(3, np.int64)).dtype == np.int32
    assert as_samples(np.zeros(3, np.float64)).dtype == np.int32          # data.py:24: every tile is cast through int32 anyway
    assert as_samples(np.zeros(3, ">u2")).dtype == np.dtype("<u2")


def test_flip_flags_are_the_draws_of_the_host_transforms():
    """FlipAugment / BatchAugment(flips only) hand the device feed (h, v) per image: applying them must equal calling the transform"""
    from unet_amd import augment as A
    from unet_amd.learner import FlipAugment
    g = torch.Generator().manual_seed(0)
    x = torch.rand(9, 2, 6, 5, generator=g)
    y = torch.randint(0, 3, (9, 6, 5), generator=g)

    def apply(flags):
        xo, yo = x.clone(), y.clone()
        for i, (h, v) in enumerate(flags):
            if h:
                xo[i], yo[i] = xo[i].flip(-1), yo[i].flip(-1)
            if v:
                xo[i], yo[i] = xo[i].flip(-2), yo[i].flip(-2)
        return xo, yo
    for mk in (lambda: FlipAugment(n_transform_imgs=0.3, seed=4),
               lambda: A.BatchAugment(A.default_pipeline(), n_transform_imgs=0.3, seed=4),
               lambda: A.BatchAugment(A.Compose([A.VerticalFlip(p=0.7), A.HorizontalFlip(p=0.2), A.VerticalFlip(p=0.5)], p=0.8), 0.5, seed=6)):
        a, b = mk(), mk()
        for _ in range(5):
            xa, ya = a(x.clone(), y.clone())
            xb, yb = apply(b.flip_flags(9))
            assert torch.equal(xa, xb) and torch.equal(ya, yb)
    assert FlipAugment(n_transform_imgs=1.0).flip_flags(8) == [(False, False)] * 8          # quirk Q7: the shipped default flips nothing
    assert not hasattr(A.BatchAugment(A.Compose([A.CoarseDropout()])), "flip_flags")


def test_gil_switch_interval_is_opt_in_and_restored(monkeypatch):
    import sys
    before = sys.getswitchinterval()
    batches = [list(range(b * 4, b * 4 + 4)) for b in range(6)]
    f = BatchFeeder(_load_factory(24), bs=4, device="cpu", depth=2, workers=2)
    monkeypatch.delenv("UNET_FEED_SWITCH_S", raising=False)
    for slot in f.run(batches):             # unset: the interpreter's setting is left alone
        assert sys.getswitchinterval() == before
        slot.release()
    monkeypatch.setenv("UNET_FEED_SWITCH_S", str(before / 10))
    inside = []
    for slot in f.run(batches):
        inside.append(sys.getswitchinterval())
        slot.release()
    assert inside and all(abs(v - before / 10) < 1e-9 for v in inside) and sys.getswitchinterval() == before
    for slot in f.run(batches):             # early exit restores it as well
        break
    assert sys.getswitchinterval() == before
    monkeypatch.setenv("UNET_FEED_SWITCH_S", str(before * 10))      # never raised above the interpreter's own
    for slot in f.run(batches):
        assert sys.getswitchinterval() == before
        slot.release()
    bad = BatchFeeder(_load_factory(24, fail_at=5), bs=4, device="cpu", depth=2, workers=2)
    monkeypatch.setenv("UNET_FEED_SWITCH_S", str(before / 10))
    with pytest.raises(OSError):
        list(bad.run(batches))
    assert sys.getswitchinterval() == before
    f.close(); bad.close()
