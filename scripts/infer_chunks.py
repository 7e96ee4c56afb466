import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tinynerf_amd import rays
from tinynerf_amd.run import TrainConfig, Trainer
dev = torch.device("cuda", 0)
o, d, rgbs, K, _ = rays.synthetic_scene(n_views=2, res=800, seed=0, device=str(dev))
for method in ("kplanes", "vanilla"):
    cfg = TrainConfig(method=method, scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
    tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev)
    lin = torch.linspace(-1, 1, 128, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    img_o, img_d = o[:640000], d[:640000]
    ref = None
    for bs in (1024, 8192, 65536, 640000):
        tr.render_rays(img_o[:bs], img_d[:bs], bs)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = tr.render_rays(img_o, img_d, bs)
        torch.cuda.synchronize(); t = time.perf_counter() - t
        if ref is None: ref = out
        print(method, "chunk", bs, f"{t*1e3:.1f} ms per 800x800 image", "max diff vs chunk 1024:", float((out - ref).abs().max()), f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
