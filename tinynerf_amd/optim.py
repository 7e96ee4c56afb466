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
        self._gated_count = {}       # device -> int32 [1]: updates that were not gated away (the bias-correction count)
        self._gated_params = {}      # device -> the parameters that share that count

    @torch.no_grad()
    def step(self, closure=None, plane_reg=None, gate=None, only=None):
        """``gate`` (harness): device float scalar; tensors outside ``plane_reg`` are only updated -- and their step count only
        advances -- when it is > 0 (tn_adam_multi_gated).  This is what torch.optim.Adam does with the ``grad is None`` parameters
        of the reference's "Empty iteration" (core.py:251-254), decided on the device instead of by a host read-back.

        ``plane_reg`` (harness): dict(spec=[(plane, H, W, C, cy, cx, cl1)], upstream=float, sums=fp64 tensor or None[, rows={id(plane):
        (row0, row1)}: the sharded pass of N > 1 -- only these rows are updated and summed, the caller all-gathers them]) folds
        the K-Planes regulariser's gradient (and its sums) into the update of those planes (tn_adam_reg_multi): the planes are
        streamed once per step.  Their new values are written to a second buffer which then becomes ``plane.data``.

        ``only`` (harness): "reg" = the ``plane_reg`` tensors alone, "rest" = everything else (the trainer may run the planes' pass on a
        stream of its own as soon as their gradients are final, beside the heads' weight-gradient kernels)."""
        loss = closure() if closure is not None else None
        reg = {}
        if plane_reg is not None:
            for slot, (p, H, W, Cc, cy, cx, cl) in enumerate(plane_reg["spec"]):
                reg[id(p)] = (slot, H, W, Cc, cy, cx, cl)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("tinynerf_amd.FusedAdam: parameters must be CUDA (HIP) tensors -- there is no CPU path")
                if only is not None and (id(p) in reg) != (only == "reg"):
                    continue
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
                # gated tensors share ONE device-side count (they are all skipped or all updated together)
                by_step.setdefault((st["step"] if (gate is None or id(p) in reg) else -1, p.device, id(p) in reg), []).append((p, g, m, v))
            # one launch per (step count, device): every tensor of the harness shares both
            for (t, dev, with_reg), tensors in by_step.items():
                common = (C.c_float(group["lr"]), C.c_float(b1), C.c_float(b2), C.c_float(group["eps"]), C.c_float(group["weight_decay"]),
                          C.c_int32(t), C.c_int32(1 if self.zero_grad_in_step else 0))
                if not with_reg:
                    items = (L.AdamItem * len(tensors))()
                    for it, (p, g, m, v) in zip(items, tensors):
                        it.param, it.grad, it.exp_avg, it.exp_avg_sq, it.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    if gate is None:
                        L.call("tn_adam_multi", dev, items, C.c_int32(len(tensors)), *common)
                    else:
                        if self._gated_count.get(dev) is None:      # device-side count of the updates that were not gated away
                            first = min(self.state[p]["step"] for p, _, _, _ in tensors) - 1
                            # [0]: the count; [1]: raised by the kernel when an updated parameter is not finite (nonfinite_flag)
                            self._gated_count[dev] = torch.tensor([first, 0], dtype=torch.int32, device=dev)
                        self._gated_params[dev] = [p for p, _, _, _ in tensors]
                        L.call("tn_adam_multi_gated", dev, items, C.c_int32(len(tensors)), *common[:5], L.ptr(self._gated_count[dev]),
                               L.ptr(gate), C.c_int32(common[6].value | 2))
                    continue
                items = (L.AdamRegItem * len(tensors))()
                for it, (p, g, m, v) in zip(items, tensors):
                    st = self.state[p]
                    if "shadow" not in st:
                        st["shadow"] = torch.empty_like(p, memory_format=torch.preserve_format)
                    slot, H, W, Cc, cy, cx, cl = reg[id(p)]
                    if (Cc, H, W) != tuple(p.shape[1:]) or not p.is_contiguous(memory_format=torch.channels_last):
                        raise RuntimeError("tinynerf_amd.FusedAdam: plane_reg expects channels_last [1,C,H,W] planes")
                    it.param, it.param_out, it.grad = p.data_ptr(), st["shadow"].data_ptr(), g.data_ptr()
                    it.exp_avg, it.exp_avg_sq, it.n = m.data_ptr(), v.data_ptr(), p.numel()
                    it.H, it.W, it.C, it.sum_slot, it.cy, it.cx, it.cl1 = H, W, Cc, slot, cy, cx, cl
                    it.row0, it.row1 = (plane_reg.get("rows") or {}).get(id(p), (0, 0))       # sharded pass: this rank's rows only
                sums = plane_reg.get("sums")
                L.call("tn_adam_reg_multi", dev, items, C.c_int32(len(tensors)), *common, C.c_float(plane_reg["upstream"]), L.ptr(sums))
                for p, _, _, _ in tensors:                   # the updated values live in the second buffer: swap
                    st = self.state[p]
                    new, st["shadow"] = st["shadow"], p.data
                    p.data = new
        return loss

    def sync_step_counts(self) -> None:
        """Gated mode keeps the real update count on the device (a skipped "Empty iteration" must not advance the bias
        correction, and deciding that on the host would be a read-back per step); ``state[p]["step"]`` counts calls.  This
        writes the device count back into the state (one 4-byte read-back), so that checkpoints, resumes and inspections see the
        count Adam actually uses."""
        for dev, cnt in self._gated_count.items():
            if cnt is None:
                continue
            c = int(cnt[0].item())
            for p in self._gated_params.get(dev, []):
                if p in self.state:
                    self.state[p]["step"] = c

    def nonfinite_flag(self, device) -> "torch.Tensor | None":
        """int32 [1] on `device`: 1 once the gated update has written a non-finite parameter (None before the first gated step).  The
        reference's loss is NaN from that step on (torch.relu hands a NaN on, models.py:7-28; the kernels' v_max_f32 does not):
        run.Trainer.loss_device() folds the flag into the loss it reports."""
        for dev, cnt in self._gated_count.items():            # ("cuda" and "cuda:0" are different dictionary keys)
            if cnt is not None and dev.type == torch.device(device).type and (torch.device(device).index in (None, dev.index)):
                return cnt[1:2]
        return None

    partial_state_reason = None        # set by run.Trainer under TrainConfig.sharded_optimizer: state_dict() would be incomplete

    def preallocate(self, params, shadow: bool = False):
        """create the Adam moments (and, `shadow`, the double buffer of the regularised pass) of `params` now, on the current stream,
        instead of at their first step -- which the harness may run on a side stream (run.Trainer, TN_ADAM_OVERLAP)"""
        for p in params:
            st = self.state[p]
            if "exp_avg" not in st:
                st.setdefault("step", 0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if shadow and "shadow" not in st:
                st["shadow"] = torch.empty_like(p, memory_format=torch.preserve_format)

    def state_dict(self):
        if self.partial_state_reason:
            raise RuntimeError("tinynerf_amd.FusedAdam.state_dict(): " + self.partial_state_reason)
        self.sync_step_counts()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._gated_count, self._gated_params = {}, {}       # re-derived from the loaded step counts at the next gated step
