"""One training step of the hot path: forward -> weighted CE -> backward -> (tile-DDP all-reduce) -> fastai Adam.

This is what ``Learner._do_one_batch`` + ``opt.step()`` execute per batch inside ``learn.fit_one_cycle``
(reference ``train.py:247-250``), as one stream of C-ABI launches with no host synchronisation.
"""
from __future__ import annotations

from typing import Optional

import torch

from .distributed import GradReducer
from .model import HipDynamicUnet
from .optimizer import FlatAdam


class TrainStep:
    def __init__(self, model: HipDynamicUnet, opt: FlatAdam, class_weights: Optional[torch.Tensor] = None, world: int = 1,
                 max_bucket_elems: int = 16 << 20):
        self.model, self.opt, self.world = model, opt, world
        self.weights = class_weights
        self.reducer: Optional[GradReducer] = None
        if world > 1:
            bounds = [model._decoder_offset] + list(model._enc_child_offset.values())
            self.reducer = GradReducer(model.flat_grad, bounds, max_bucket_elems)
            model.grad_ready_hook = self.reducer.ready_down_to

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """x [B,C,H,W] fp32, y [B,H,W] int64 (device tensors).  Returns the (rank-local) loss as a device scalar."""
        if self.reducer is not None:
            self.reducer.reset()
        loss = self.model.forward_loss_backward(x, y, self.weights, grad_scale=1.0 / self.world)
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return loss
