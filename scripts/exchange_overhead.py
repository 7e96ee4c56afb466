#!/usr/bin/env python3
"""What the N > 1 step costs apart from the wire: bench.py's K-Planes workload under a one-rank RCCL group, Trainer told
world_size = 1 and 2 (every collective of the exchange path issued over one rank: the sums are identities, the launches,
stream hand-offs and RCCL kernels are real)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", TORCH_NCCL_HIGH_PRIORITY="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from tinynerf_amd import rays                                    # noqa: E402
from tinynerf_amd.run import TrainConfig, Trainer                # noqa: E402

o, d, rgbs, K, _ = rays.synthetic_scene(n_views=8, res=800, seed=0, device=str(dev))
for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 1, 2]):
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
    tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev, rank=0, world_size=world)
    lin = torch.linspace(-1, 1, 128, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = sum(tr.step()["n_samples"] for _ in range(40))
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    rows = tr._reduce_rows
    print(f"world_size {world}: {t / 40 * 1e3:.3f} ms per step, {n / t:.4g} samples/s" + (f", live rows per plane {rows}" if rows else ""))
    del tr
torch.distributed.destroy_process_group()
