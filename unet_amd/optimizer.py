"""fastai Adam restated over the flat parameter buffer, one fused HIP launch per step.

Replaces ``opt_func=Adam`` (``train.py:218``): fastai ``Adam(mom=.9, sqr_mom=.99, eps=1e-5,
wd=0.01, decouple_wd=True)`` = weight_decay -> average_grad(dampening) -> average_sqr_grad ->
step_stat -> adam_step (optimizer.py), with the three parameter groups of ``_xresnet_split``
(``train.py:78-80``) and no weight decay on norm / bias parameters (``wd_bn_bias=False``,
``train.py:102,152-154``).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops


def xresnet_split(model) -> List[List[nn.Parameter]]:
    """stem m[0][:3] / rest of encoder m[0][3:] / decoder m[1:]"""
    enc = model[0]
    g0 = [p for l in list(enc)[:3] for p in l.parameters()]
    g1 = [p for l in list(enc)[3:] for p in l.parameters()]
    g2 = [p for l in list(model.layers)[1:] for p in l.parameters()]
    return [g0, g1, g2]


def norm_bias_params(model) -> List[nn.Parameter]:
    out = []
    for l in model.modules():
        if isinstance(l, (nn.BatchNorm1d, nn.BatchNorm2d)):
            out += list(l.parameters(recurse=False))
        elif isinstance(getattr(l, "bias", None), nn.Parameter):
            out.append(l.bias)
    return out


class FlatAdam:
    """State (grad_avg, sqr_avg) lives in two flat buffers parallel to model.flat_param."""

    def __init__(self, model, lr=1e-3, mom=0.9, sqr_mom=0.99, eps=1e-5, wd=0.01, wd_bn_bias=False, splitter=xresnet_split):
        self.model = model
        self.groups = splitter(model)
        assert len(self.groups) <= 4
        self.mom, self.sqr_mom, self.eps, self.wd = mom, sqr_mom, eps, wd
        self.set_lr(lr)
        n = model.flat_param.numel()
        dev = model.flat_param.device
        self.grad_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.sqr_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        code = np.zeros(n, dtype=np.uint8)
        no_wd = set() if wd_bn_bias else {id(p) for p in norm_bias_params(model)}
        seen = set()
        for gi, g in enumerate(self.groups):
            for p in g:
                o, cnt = model.param_span(p)
                code[o:o + cnt] = gi | (0 if id(p) in no_wd else 4)
                seen.add(id(p))
        missing = [n_ for n_, p in model.named_parameters() if id(p) not in seen]
        assert not missing, f"parameters outside every group: {missing[:4]}"
        self.code = torch.from_numpy(code).to(dev)
        self.step_count = 0
        self.grad_scale = 1.0
        # hipGraph mode: hyper-parameters live in a device block refreshed by the host before each replay
        nh = ops.lib.unet_adam_hyper_floats()
        # a RING of pinned staging blocks: the host runs ahead of the stream, so block i may only be refilled once the
        # H2D copy that read it has executed (event recorded right behind that copy)
        self._hyper_ring = [torch.zeros(nh, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(nh)
                            for _ in range(4)]
        self._hyper_events = [None] * len(self._hyper_ring)
        self._hyper_slot = 0
        self._hyper_dev = torch.zeros(nh, dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        k = len(self.groups)
        self.lrs = [float(lr)] * k if np.isscalar(lr) else [float(v) for v in lr]
        assert len(self.lrs) == k

    def step(self):
        self.step_count += 1
        m = self.model
        ops.adam_step(m.flat_param, m.flat_grad, self.grad_avg, self.sqr_avg, self.code, self.lrs, self.mom, self.sqr_mom,
                      self.eps, self.wd, self.step_count, self.grad_scale)
        m.mark_weights_dirty()

    def upload_hyper(self, step: int):
        """host -> device copy of (lr, momentum, de-bias terms ...) for step number `step` (stream ordered, no sync)"""
        import ctypes as C
        arr = (C.c_float * 4)(*(self.lrs + [0.0] * (4 - len(self.lrs))))
        i = self._hyper_slot
        self._hyper_slot = (i + 1) % len(self._hyper_ring)
        if self._hyper_events[i] is not None:
            self._hyper_events[i].synchronize()          # the copy that last read this block is done
        host = self._hyper_ring[i]
        hp = C.cast(host.data_ptr(), C.POINTER(C.c_float))
        ops.check(ops.lib.unet_adam_fill_hyper(hp, arr, float(self.mom), float(self.sqr_mom), float(self.eps), float(self.wd), int(step),
                                               float(self.grad_scale)), "adam_fill_hyper")
        self._hyper_dev.copy_(host, non_blocking=True)
        if self._hyper_dev.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_events[i] = ev

    def step_from_device_hyper(self):
        """the launch that gets captured in a hipGraph: every hyper-parameter is read from self._hyper_dev"""
        m = self.model
        ops.check(ops.lib.unet_adam_step_dev(m.flat_param.data_ptr(), m.flat_grad.data_ptr(), self.grad_avg.data_ptr(),
                                             self.sqr_avg.data_ptr(), self.code.data_ptr(), m.flat_param.numel(),
                                             self._hyper_dev.data_ptr(), ops._stream()), "adam_step_dev")
        m.mark_weights_dirty()

    def zero_grad(self):
        pass  # every backward overwrites the flat gradient buffer

    def state_dict(self):
        return {"grad_avg": self.grad_avg, "sqr_avg": self.sqr_avg, "step": self.step_count, "lrs": self.lrs, "mom": self.mom}

    def load_state_dict(self, sd):
        self.grad_avg.copy_(sd["grad_avg"]); self.sqr_avg.copy_(sd["sqr_avg"])
        self.step_count = int(sd["step"]); self.lrs = list(sd["lrs"]); self.mom = float(sd["mom"])
