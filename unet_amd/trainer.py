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
    """use_graph=True (single GPU): after two eager warm-up calls the whole step -- ~700 launches -- is captured once in a
    hipGraph (torch.cuda.CUDAGraph on the launch stream) and replayed; inputs are copied into static buffers and the Adam
    hyper-parameters (lr, momentum, step count) are refreshed in a device block before each replay.  Pays when the step is
    launch-bound (small tiles / small batches: BASELINE configs[0]); at batch 16 of 512x512 tiles the GPU is never idle."""

    def __init__(self, model: HipDynamicUnet, opt: FlatAdam, class_weights: Optional[torch.Tensor] = None, world: int = 1,
                 max_bucket_elems: int = 16 << 20, use_graph: bool = False):
        self.model, self.opt, self.world = model, opt, world
        self.weights = class_weights
        self.reg_kind: Optional[str] = None     # "mse" | "l1" | "smoothl1": regression mode (float targets, n_out = 1)
        self.reg_beta = 0.5
        self.focal_gamma: Optional[float] = None    # FocalLossFlat(gamma) instead of the weighted cross-entropy (params_and_main.py:87-89)
        self.use_graph = use_graph and world == 1
        if self.use_graph:
            from .modules import SelfAttention
            if any(isinstance(mod, SelfAttention) for mod in model.modules()):
                # the spectral-norm power iteration and its adjoint are torch autograd inside the step: capturing them ends in a segmentation
                # fault of hipStreamEndCapture on this stack (scripts/graph_sa_probe.py).  Eager is what the measurements use anyway
                # (a replayed graph is not faster: DESIGN 6)
                import warnings
                warnings.warn("TrainStep(use_graph=True): models with SelfAttention run eagerly (torch autograd inside the step is not capturable here)")
                self.use_graph = False
        if self.use_graph:
            # a replayed graph has no host in the loop: the second stream of the weight gradients buys nothing there and its
            # fork / join nodes cost (cfg1 fp32: 7.51 ms per replay without, 7.69 with; bf16: 4.00 / 4.33)
            model.ctx.wgrad_overlap = False
        self._graph = None
        self._calls = 0
        self._xs = self._ys = self._loss = None
        self.reducer: Optional[GradReducer] = None
        # bench.py: a list here collects (start, end) events around the point where the compute stream waits for the gradient buckets
        self.comm_events: Optional[list] = None
        if world > 1:
            bounds = [model._decoder_offset] + list(model._layer_offset.values()) + list(model._enc_child_offset.values())
            self.reducer = GradReducer(model.flat_grad, bounds, max_bucket_elems)
            model.grad_ready_hook = self.reducer.ready_down_to

    def _graphed(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        m, opt = self.model, self.opt
        if self._graph is not None and (tuple(x.shape) != tuple(self._xs.shape)):
            self._graph = None                      # new geometry: re-capture
            self._calls = 0
        if self._graph is None:
            if self._calls < 2:                     # eager warm-up: allocates every persistent buffer / workspace
                self._calls += 1
                loss = m.forward_loss_backward(x, y, self.weights, grad_scale=1.0, reg_kind=self.reg_kind, reg_beta=self.reg_beta, focal_gamma=self.focal_gamma)
                opt.step()
                return loss
            self._xs = x.to(m._device, torch.float32).clone()
            self._ys = y.to(m._device, torch.int64 if self.reg_kind is None else torch.float32).clone()
            opt.upload_hyper(opt.step_count + 1)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._loss = m.forward_loss_backward(self._xs, self._ys, self.weights, grad_scale=1.0, reg_kind=self.reg_kind,
                                                     reg_beta=self.reg_beta, focal_gamma=self.focal_gamma)
                opt.step_from_device_hyper()
            # (the capture itself does not execute the step)
        self._xs.copy_(x, non_blocking=True)
        self._ys.copy_(y, non_blocking=True)
        opt.step_count += 1
        opt.upload_hyper(opt.step_count)
        self._graph.replay()
        # every replay re-packs the filters from the pre-step parameters and then runs Adam: the packed images a later
        # eager (eval / predict) forward would reuse are one step stale -> invalidate them after EVERY replay
        m.mark_weights_dirty()
        return self._loss

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """x [B,C,H,W] fp32, y [B,H,W] int64 (device tensors).  Returns the (rank-local) loss as a device scalar."""
        if self.use_graph:
            return self._graphed(x, y)
        if self.reducer is not None:
            self.reducer.reset()
        loss = self.model.forward_loss_backward(x, y, self.weights, grad_scale=1.0, reg_kind=self.reg_kind, reg_beta=self.reg_beta,
                                                world=self.world, focal_gamma=self.focal_gamma)
        if self.reducer is not None:
            ev = self.comm_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.reducer.finish()
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        self.opt.step()
        return loss
