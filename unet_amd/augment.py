"""On-device batch augmentation with the semantics of the reference's ``SegmentationAlbumentationsTransform`` (``utils.py:170-295``)
and of the albumentations transforms its configuration names (``params_and_main.py:105-115``: ``A.Compose([A.HorizontalFlip(p=0.5),
A.VerticalFlip(p=0.5), # A.RandomBrightnessContrast(...), # A.CoarseDropout(p=0.5)])``).  albumentations is not installed here and
its transforms run on the host per image; these run on the GPU on the already scaled float batch.  A pipeline written for the
reference ports by changing the import: ``from unet_amd import augment as A``.

Kept quirks: only the FIRST ``ceil(B * n_transform_imgs) - B`` images of a batch are candidates (python slice semantics of
``utils.py:255-256``), so the shipped default ``n_transform_imgs = 1`` augments NOTHING (quirk Q7).  The reference augments the
image in [0, 1] (``img / 255`` for int8 data, ``utils.py:262-265``) -- the same domain as the batches here.  Random draws come from
a seeded numpy generator (albumentations uses python's ``random``): the streams differ, the distributions are the same.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch


class _Transform:
    def __init__(self, p: float = 0.5, always_apply: bool = False):
        self.p = 1.0 if always_apply else float(p)

    def apply(self, img: torch.Tensor, mask: torch.Tensor, g: np.random.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def __call__(self, img, mask, g):
        return self.apply(img, mask, g) if g.random() < self.p else (img, mask)


class HorizontalFlip(_Transform):
    """albumentations ``HorizontalFlip``: image [C,H,W] and mask [H,W] mirrored along the width"""
    def apply(self, img, mask, g):
        return img.flip(-1), mask.flip(-1)


class VerticalFlip(_Transform):
    def apply(self, img, mask, g):
        return img.flip(-2), mask.flip(-2)


class RandomBrightnessContrast(_Transform):
    """albumentations ``RandomBrightnessContrast(brightness_limit, contrast_limit, brightness_by_max=True, p)`` on a float image in
    [0, 1]: ``img * alpha + beta * (1 if brightness_by_max else mean(img))`` with alpha = 1 + U(contrast_limit), beta = U(brightness_limit),
    clipped to [0, 1]; the mask is untouched."""

    def __init__(self, brightness_limit=0.2, contrast_limit=0.2, brightness_by_max=True, p=0.5, always_apply=False):
        super().__init__(p, always_apply)
        lim = lambda v: (-abs(v), abs(v)) if np.isscalar(v) else (float(v[0]), float(v[1]))
        self.b, self.c, self.by_max = lim(brightness_limit), lim(contrast_limit), brightness_by_max

    def apply(self, img, mask, g):
        alpha = 1.0 + g.uniform(*self.c)
        beta = g.uniform(*self.b)
        out = img * alpha
        if beta != 0:
            out = out + (beta if self.by_max else beta * img.mean())
        return out.clamp_(0.0, 1.0), mask


class CoarseDropout(_Transform):
    """albumentations ``CoarseDropout(max_holes=8, max_height=8, max_width=8, min_holes=None, min_height=None, min_width=None,
    fill_value=0, mask_fill_value=None, p)``: between min_holes and max_holes rectangles of the image set to fill_value (the mask
    only when mask_fill_value is given); the unset minima default to the maxima."""

    def __init__(self, max_holes=8, max_height=8, max_width=8, min_holes=None, min_height=None, min_width=None, fill_value=0,
                 mask_fill_value=None, p=0.5, always_apply=False):
        super().__init__(p, always_apply)
        self.holes = (max_holes if min_holes is None else min_holes, max_holes)
        self.h = (max_height if min_height is None else min_height, max_height)
        self.w = (max_width if min_width is None else min_width, max_width)
        self.fill, self.mask_fill = fill_value, mask_fill_value

    def apply(self, img, mask, g):
        H, W = img.shape[-2:]
        img = img.clone()
        mask = mask if self.mask_fill is None else mask.clone()
        for _ in range(int(g.integers(self.holes[0], self.holes[1] + 1))):
            hh, ww = int(g.integers(self.h[0], self.h[1] + 1)), int(g.integers(self.w[0], self.w[1] + 1))
            hh, ww = min(hh, H), min(ww, W)
            y1, x1 = int(g.integers(0, H - hh + 1)), int(g.integers(0, W - ww + 1))
            img[..., y1:y1 + hh, x1:x1 + ww] = self.fill
            if self.mask_fill is not None:
                mask[y1:y1 + hh, x1:x1 + ww] = self.mask_fill
        return img, mask


class Compose:
    """albumentations ``Compose``: the transforms in order, each with its own probability; ``p`` gates the whole pipeline"""

    def __init__(self, transforms: Sequence[_Transform], p: float = 1.0):
        self.transforms: List[_Transform] = list(transforms)
        self.p = float(p)

    def __call__(self, img, mask, g):
        if g.random() >= self.p:
            return img, mask
        for t in self.transforms:
            img, mask = t(img, mask, g)
        return img, mask


class BatchAugment:
    """``SegmentationAlbumentationsTransform.encodes`` on a device batch: pipeline ``aug`` on the first ``ceil(B * n_transform_imgs) - B``
    images (``utils.py:239-291``), the rest unchanged."""

    def __init__(self, aug: Compose, n_transform_imgs: float = 1.0, seed: int = 0):
        if not (0 <= n_transform_imgs <= 1):
            raise ValueError(f"The n_transform_imgs parameter ({n_transform_imgs}) must be between 1 and 0.")       # utils.py:235-237
        self.aug, self.n, self.g = aug, n_transform_imgs, np.random.default_rng(seed)

    def __call__(self, xb: torch.Tensor, yb: torch.Tensor):
        B = xb.shape[0]
        n_transform = math.ceil(B * self.n)
        for i in list(range(B))[:n_transform - B]:
            xi, yi = self.aug(xb[i], yb[i], self.g)
            xb[i], yb[i] = xi, yi
        return xb, yb

    @property
    def flips_only(self) -> bool:
        return all(type(t) in (HorizontalFlip, VerticalFlip) for t in self.aug.transforms)

    def __getattr__(self, name):
        # `flip_flags` exists only for pipelines made of flips (the reference's default, params_and_main.py:105-115): the device feed
        # (learner.DataLoader) then folds the flips into its staging kernels instead of running torch ops per image
        if name == "flip_flags" and self.flips_only:
            return self._flip_flags
        raise AttributeError(name)

    def _flip_flags(self, B: int) -> list:
        """the random draws of ``__call__`` in its order, reduced to (mirror along the width, mirror along the height) per image"""
        n_transform = math.ceil(B * self.n)
        flags = [(False, False)] * B
        for i in list(range(B))[:n_transform - B]:
            h = v = False
            if self.g.random() < self.aug.p:           # Compose.__call__: `if g.random() >= self.p: return`
                for t in self.aug.transforms:          # _Transform.__call__: `if g.random() < self.p: apply`
                    if self.g.random() < t.p:
                        if type(t) is HorizontalFlip:
                            h = not h
                        else:
                            v = not v
            flags[i] = (h, v)
        return flags


def default_pipeline() -> Compose:
    """the reference's shipped ``aug_pipe`` (params_and_main.py:105-115)"""
    return Compose([HorizontalFlip(p=0.5), VerticalFlip(p=0.5)])
