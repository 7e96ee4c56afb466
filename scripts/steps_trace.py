import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tinynerf_amd import rays
from tinynerf_amd.run import TrainConfig, Trainer
method = sys.argv[1]
dev = torch.device("cuda", 0)
o, d, rgbs, K, _ = rays.synthetic_scene(n_views=8, res=800, seed=0, device=str(dev))
cfg = TrainConfig(method=method, scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev)
lin = torch.linspace(-1, 1, 128, device=dev)
zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
tr.occupancy_grid_updates = 10 ** 9
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    torch.cuda.synchronize(); t = time.perf_counter()
    st = tr.step()
    torch.cuda.synchronize(); t = time.perf_counter() - t
    print(i, f"{t*1e3:.2f} ms", int(st["n_samples"]), int(st["n_rays"]), f"alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
