"""K-Planes convergence over seeds with both forms of the heads' forward (f16x2 splits / fp32 MFMA): held-out PSNR at steps 0, 500, ..., 2000 on the
scene of scripts/convergence.py.  usage (GPU box): python scripts/convergence_seeds.py [n_seeds]   -> profiles/round4_convergence.json"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tinynerf_amd import models, rays
from tinynerf_amd.run import TrainConfig, Trainer, psnr
dev = torch.device("cuda", 0)
V, RES = 41, 100
o, d, rgbs, K, _ = rays.synthetic_scene(n_views=V, res=RES, seed=0, device=str(dev))
per = RES * RES
tr_s, he_s = slice(0, (V - 1) * per), slice((V - 1) * per, V * per)
res = {}
for mode in ("f16x2", "fp32"):
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        models.MATMUL = mode
        torch.manual_seed(seed)
        cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=1024, n_samples=256, seed=seed)
        t = Trainer(cfg, o[tr_s], d[tr_s], rgbs[tr_s], torch.ones(3, device=dev), dev)
        curve = []
        for step in range(2001):
            if step % 500 == 0:
                with torch.no_grad():
                    img = t.render_rays(o[he_s], d[he_s])
                curve.append(float(psnr(img, rgbs[he_s])))
            if step < 2000: t.step()
        res.setdefault(mode, []).append(curve)
        print(mode, seed, [round(c, 2) for c in curve], flush=True)
for mode, cs in res.items():
    a = np.array(cs)
    print(mode, "mean", np.round(a.mean(0), 2), "std", np.round(a.std(0, ddof=1), 2))
