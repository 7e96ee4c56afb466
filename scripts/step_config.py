#!/usr/bin/env python3
"""A few training steps of one model configuration (vanilla / cobafa / kplanes) on bench.py's workload, for
`scripts/kstats.sh <tag> scripts/step_config.py <method> [steps]` (per-kernel times of BASELINE configs 2 and 5)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinynerf_amd import rays                                    # noqa: E402
from tinynerf_amd.run import TrainConfig, Trainer                # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "vanilla"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if "--layerwise" in sys.argv:                                    # A / B: the wide stacks' forward passes one launch per layer (TN_MLP_LAYERWISE)
    from tinynerf_amd.models import _FusedMLP
    _FusedMLP.layerwise_training = _FusedMLP.layerwise_inference = True
dev = torch.device("cuda", 0)
o, d, rgbs, K, _ = rays.synthetic_scene(n_views=8, res=800, seed=0, device=str(dev))
cfg = TrainConfig(method=method, scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev)
lin = torch.linspace(-1, 1, 128, device=dev)
zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
tr.occupancy_grid_updates = 10 ** 9
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
t = time.perf_counter()
n = sum(tr.step()["n_samples"] for _ in range(steps))
torch.cuda.synchronize()
t = time.perf_counter() - t
print(f"{method}: {t / steps * 1e3:.3f} ms per step, {n / t:.4g} samples/s, loss {tr.loss_value():.5f}")
