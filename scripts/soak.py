import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tinynerf_amd import rays
from tinynerf_amd.run import TrainConfig, Trainer, psnr
method, steps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
o, d, rgbs, K, _ = rays.synthetic_scene(n_views=8, res=200, seed=0, device=str(dev))
cfg = TrainConfig(method=method, scene_type="aabb", batch_size=1024, n_samples=256, seed=0)
tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev)
t0 = time.perf_counter()
for i in range(steps):
    st = tr.step()
    if i % 50 == 0 or i == steps - 1:
        print(i, f"loss {tr.loss_value():.5f} samples {int(st['n_samples'])} rays {int(st['n_rays'])} occ {tr.occupancy_grid.occupancy():.3f} reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB")
torch.cuda.synchronize()
print(method, f"{(time.perf_counter()-t0)/steps*1e3:.2f} ms/step")
img = tr.render_rays(o[:40000], d[:40000])
print("psnr on view 0:", float(psnr(img, rgbs[:40000])), "finite params:", all(torch.isfinite(p).all().item() for p in tr.renderer.parameters()))
