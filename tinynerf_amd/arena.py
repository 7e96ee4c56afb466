"""Capacity-based scratch buffers for the training harness.

Dynamic batches (reference run.py:215-244) change N by a few percent every step; a caching allocator that sees a new
multi-GB workspace size at every record high of N falls back to hipMalloc in the middle of the step (measured: +60 ms
for the 2.4 GB K-Planes workspace, 100-240 ms for the 11 GB one of the width-256 stack).  Buffers are allocated once
with 25 % headroom and handed out as views.  Only valid when each forward's backward runs before the next forward of
the same module (the training loop): opt-in, the module API allocates per call.
"""
from __future__ import annotations

from typing import Dict

import torch


class Arena:
    def __init__(self) -> None:
        self.buf: Dict[str, torch.Tensor] = {}
        self.grown = 0          # how often a buffer had to be (re)allocated: flat after warm-up, asserted by bench.py

    def get(self, name: str, shape, dev: torch.device, dtype=torch.float32) -> torch.Tensor:
        numel = 1
        for d in shape:
            numel *= int(d)
        t = self.buf.get(name)
        if t is None or t.numel() < numel or t.device != dev or t.dtype != dtype:
            self.buf.pop(name, None)
            t = None                                    # release before growing
            t = torch.empty(int(numel * 1.25) + 1024, device=dev, dtype=dtype)
            self.buf[name] = t
            self.grown += 1
        return t[:numel].view(*shape)

    def reserve(self, name: str, numel: int, dev: torch.device, dtype=torch.float32) -> None:
        """size a buffer ahead of its first use (e.g. from the target sample count of the dynamic batch)"""
        t = self.buf.get(name)
        if t is None or t.numel() < numel or t.device != dev or t.dtype != dtype:
            self.buf.pop(name, None)
            self.buf[name] = torch.empty(int(numel), device=dev, dtype=dtype)
            self.grown += 1
