"""Adam with the update of ``torch.optim.Adam`` (reference run.py:186: lr 1e-2, eps 1e-15, coupled weight decay)
as ONE kernel pass over all parameter tensors (``tn_adam_multi``: 28 B/element, one launch) instead of torch's multi-kernel
foreach path (~8 passes over the 126 MiB of K-Planes planes).  ``param_groups`` behave as in torch, so
``MultiStepLR`` drives it unchanged."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Tuple

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, zero_grad_in_step: bool = False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.zero_grad_in_step = zero_grad_in_step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("tinynerf_amd.FusedAdam: parameters must be CUDA (HIP) tensors -- there is no CPU path")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                same = g.stride() == p.stride() == m.stride() == v.stride()
                dense = (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                         or (p.dim() == 5 and p.is_contiguous(memory_format=torch.channels_last_3d)))
                if not (same and dense and p.dtype == torch.float32):
                    raise RuntimeError("tinynerf_amd.FusedAdam: parameter, gradient and state must be dense fp32 with equal strides")
                by_step.setdefault((st["step"], p.device), []).append((p, g, m, v))
            # one launch per (step count, device): every tensor of the harness shares both
            for (t, dev), tensors in by_step.items():
                items = (L.AdamItem * len(tensors))()
                for it, (p, g, m, v) in zip(items, tensors):
                    it.param, it.grad, it.exp_avg, it.exp_avg_sq, it.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                L.call("tn_adam_multi", dev, items, C.c_int32(len(tensors)), C.c_float(group["lr"]), C.c_float(b1), C.c_float(b2),
                       C.c_float(group["eps"]), C.c_float(group["weight_decay"]), C.c_int32(t), C.c_int32(1 if self.zero_grad_in_step else 0))
        return loss
