#!/usr/bin/env python3
"""Capture golden input/output vectors from the imported reference.

TEST INFRASTRUCTURE ONLY -- runs in the build container (where /root/reference
exists), never on the GPU box.  It copies the reference to a scratch directory so
nothing is ever written into /root/reference, stubs ``torch.utils.cpp_extension.load``
(reference src/core.py:7 would otherwise hipify + JIT-compile src/cuda.cu) and
replaces ``_cuda`` with the C restatement ``oracle/weights_ref.c`` so that
``NerfRenderer.forward`` can run on CPU.  Outputs: ``tests/golden/G*.npz``.

Usage:  python oracle/make_goldens.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import shutil
import sys
import tempfile
import types

import numpy as np

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def import_reference(ref_dir):
    import torch
    import torch.utils.cpp_extension as ce
    import tinynerf_oracle as orc

    scratch = tempfile.mkdtemp(prefix="tinynerf_ref_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(ref_dir, dst)

    def fwd(s, d, info, thr):
        return torch.from_numpy(orc.weights_fwd(s.detach().numpy(), d.detach().numpy(), info.numpy(), float(thr)))

    def bwd(s, d, info, w, g):
        return torch.from_numpy(orc.weights_bwd(s.detach().numpy(), d.detach().numpy(), info.numpy(),
                                                w.detach().numpy(), g.detach().numpy()))

    ce.load = lambda *a, **k: types.SimpleNamespace(compute_weights_fwd=fwd, compute_weights_bwd=bwd)
    sys.path.insert(0, dst)
    import warnings
    warnings.filterwarnings("ignore")
    import src.core as core
    import src.models as models
    import src.data as data
    return torch, core, models, data, scratch


def sd_np(module):
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    args = ap.parse_args()
    torch, core, models, data, scratch = import_reference(args.ref)
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(1)
    meta = dict(torch_version=torch.__version__)

    def save(name, **kw):
        np.savez_compressed(os.path.join(out_dir, name + ".npz"),
                            **{k: (v.detach().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in kw.items()})
        print("wrote", name, {k: np.asarray(v.detach().numpy() if hasattr(v, 'detach') else v).shape for k, v in kw.items()})

    # ---- G1: RayMarcherAABB + ContractionAABB (core.py:22-31,61-88) ---------------------
    torch.manual_seed(1)
    R, S = 64, 32
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    o = torch.randn(R, 3) * 2.5
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(R, 3), dim=-1)
    d[0, 1] = 0.0                       # zero direction component (eps branch, core.py:79)
    d[1] = torch.tensor([0.0, 0.0, 1.0]); o[1] = torch.tensor([0.2, -0.3, -4.0])   # axis-aligned
    o[2] = torch.tensor([5.0, 5.0, 5.0]); d[2] = torch.tensor([0.0, 1.0, 0.0])     # misses the box
    o[3] = torch.tensor([0.1, 0.2, -0.3])                                           # starts inside
    m = core.RayMarcherAABB(aabb, S, 0.1)
    t, dl = m(o, d)
    pts = o[:, None, :] + d[:, None, :] * t[..., None]
    c, mask = core.ContractionAABB(aabb)(pts)
    save("G1_march_aabb", rays_o=o, rays_d=d, aabb=aabb, n_samples=S, near=0.1, far=1e5,
         step_size=m.step_size, t=t, delta=dl, coords=c, mask=mask)
    aabb2 = torch.tensor([[0., 0., 0.], [2., 1., 3.]])
    m2 = core.RayMarcherAABB(aabb2, 17)
    t2, dl2 = m2(o, d)
    save("G1b_march_aabb_asym", rays_o=o, rays_d=d, aabb=aabb2, n_samples=17, near=0.0, far=1e5,
         step_size=m2.step_size, t=t2, delta=dl2)

    # ---- G2: RayMarcherUnbounded + ContractionMip360 (core.py:11-20,36-59) -------------
    for S2, near, rng in [(8, 0.1, 0.9), (200, 0.0, 1.0), (1000, 0.1, 1.7)]:
        mu = core.RayMarcherUnbounded(S2, near, 1e5, rng)
        t, dl = mu(o, d)
        nr = 64 if S2 <= 8 else 6
        pts = o[:nr, None, :] + d[:nr, None, :] * t[:nr, :, None]
        c_inf, _ = core.ContractionMip360(float("inf"))(pts)
        c_2, _ = core.ContractionMip360(2)(pts)
        save(f"G2_unbounded_S{S2}", rays_o=o[:nr], rays_d=d[:nr], n_samples=S2, near=near, uniform_range=rng,
             t_row=t[0], delta_row=dl[0], step_size=mu.step_size, coords_inf=c_inf, coords_l2=c_2)

    # ---- G3: OccupancyGrid.forward (core.py:147-156) ------------------------------------
    torch.manual_seed(3)
    g = core.OccupancyGrid([12, 20, 16], 1 / 1024.)
    decay = 0.01 ** (1 / 16)
    k = torch.randint(0, 24, (12, 20, 16))
    g.grid.copy_(torch.tensor(decay) ** k.float())
    g.mean = g.grid.mean().item()
    pts = torch.rand(4096, 3) * 2.6 - 1.3
    pts[:64] = torch.tensor([-1., 1., 0., 0.5])[torch.randint(0, 4, (64, 3))]
    vox = torch.stack(torch.meshgrid(torch.arange(16.), torch.arange(20.), torch.arange(12.), indexing="ij"), -1).view(-1, 3)
    on_vox = 2. * vox / torch.tensor([15., 19., 11.]) - 1.      # exactly on voxel centres
    pts = torch.cat([pts, on_vox[:512]])
    occ = g(pts)
    vals = torch.nn.functional.grid_sample(g.grid[None, None], pts.view(1, -1, 1, 1, 3), align_corners=True).view(-1)
    save("G3_occupancy_query", grid=g.grid, coords=pts, threshold=g.threshold, occupied=occ, values=vals)
    # the reference's own known-answer test (tests/test_core.py:5-38)
    g = core.OccupancyGrid(128, 1 / 1024.)
    g.grid[:, :, 64:] = 0.
    cs = torch.tensor([[32, 32, 32], [32, 32, 96], [32, 96, 32], [32, 96, 96], [96, 32, 32], [96, 32, 96], [96, 96, 32], [96, 96, 96]])
    unit = 2. * (cs / g.size) - 1.
    save("G3b_reference_known_answer", coords=unit, occupied=g(unit), threshold=g.threshold)

    # ---- G4: RayProvider (core.py:158-188) ------------------------------------------------
    torch.manual_seed(4)
    R, S = 96, 48
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    o = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1) * 4.0311
    d = torch.nn.functional.normalize(-o + 0.6 * torch.randn(R, 3), dim=-1)
    g = core.OccupancyGrid(32, 1 / 1024.)
    kk = torch.randint(0, 30, (32, 32, 32))
    g.grid.copy_(torch.tensor(decay) ** kk.float())
    g.grid[:, :, 20:] = 0.
    g.mean = g.grid.mean().item()
    m = core.RayMarcherAABB(aabb, S, 0.1)
    rp = core.RayProvider(g, core.ContractionAABB(aabb), m)
    packed, info = rp(o, d, training=False)
    torch.manual_seed(44)
    jit = torch.rand(R, S)
    torch.manual_seed(44)
    packed_j, info_j = rp(o, d, training=True)
    save("G4_ray_provider_aabb", rays_o=o, rays_d=d, aabb=aabb, n_samples=S, near=0.1, grid=g.grid,
         threshold=g.threshold, packed=packed, info=info, jitter=jit, packed_jit=packed_j, info_jit=info_j)
    mu = core.RayMarcherUnbounded(S, 0.1, 1e5, 1.3)
    rpu = core.RayProvider(g, core.ContractionMip360(float("inf")), mu)
    o2 = torch.rand(R, 3) - 0.5
    packed_u, info_u = rpu(o2, d, training=False)
    torch.manual_seed(45)
    jit_u = torch.rand(R, S)
    torch.manual_seed(45)
    packed_uj, info_uj = rpu(o2, d, training=True)
    save("G4b_ray_provider_unbounded", rays_o=o2, rays_d=d, n_samples=S, near=0.1, uniform_range=1.3,
         grid=g.grid, threshold=g.threshold, packed=packed_u, info=info_u, jitter=jit_u,
         packed_jit=packed_uj, info_jit=info_uj)

    # ---- G5: OccupancyGrid.update (core.py:133-145) ----------------------------------------
    torch.manual_seed(5)
    fm = models.VanillaFeatureMLP(4, 32, 2)
    od = models.VanillaOpacityDecoder(32)
    with torch.no_grad():                    # make alpha straddle the 0.01 threshold with spatial variation
        od.net.net[2].weight.mul_(6.0)
        probe = torch.rand(4096, 3) * 2 - 1
        y = od.net(fm(probe))
        od.net.net[2].bias.add_(-0.6 - y.median())
    sigma_fn = lambda x: od(fm(x))
    g = core.OccupancyGrid([16, 12, 20], 0.05, 0.01, decay)
    torch.manual_seed(55)
    jitters = torch.stack([torch.rand(12, 20, 3) for _ in range(16)])
    torch.manual_seed(55)
    g.update(sigma_fn)
    grid1, mean1 = g.grid.clone(), g.mean
    assert 0.1 < (grid1 == 1).float().mean() < 0.9, (grid1 == 1).float().mean()
    g.update(sigma_fn)            # second sweep continues the RNG stream
    save("G5_occupancy_update", jitters=jitters, step_size=0.05, base_threshold=0.01, decay=decay,
         grid_after_1=grid1, mean_after_1=mean1, occupancy_after_1=float((grid1 > min(0.01, mean1)).sum().item() / grid1.numel()),
         **{"fm." + k: v for k, v in sd_np(fm).items()}, **{"od." + k: v for k, v in sd_np(od).items()})

    # ---- G6: PositionalEncoding (models.py:30-39) -----------------------------------------
    torch.manual_seed(6)
    x = torch.rand(64, 3) * 2 - 1
    save("G6_posenc", x=x, **{f"enc{F}": models.PositionalEncoding(F)(x) for F in (3, 8, 10)},
         **{f"freqs{F}": models.PositionalEncoding(F).freqs for F in (3, 8, 10)},
         x4=torch.rand(2, 3, 5, 3), enc4=None or models.PositionalEncoding(4)(torch.zeros(2, 3, 5, 3)).shape)

    # ---- G7: K-Planes plane + field (models.py:93-163) -------------------------------------
    pl = models.KPlanesFeaturePlane(1, (3, 5))
    with torch.no_grad():
        pl.plane.copy_(torch.arange(15.).view(1, 1, 3, 5))
    probe = torch.tensor([[-1., -1.], [1., -1.], [-1., 1.], [1., 1.], [0., 0.], [0.3, -0.7], [1.2, 0.1], [-1.0, 1.0001]])
    save("G7a_plane_arange", plane=pl.plane, xy=probe, out=pl(probe).view(-1, 1))
    torch.manual_seed(7)
    field = models.KPlanesFeatureField(32)
    res = [(8, 8), (12, 10), (16, 16)]
    field.planes = torch.nn.ModuleList([torch.nn.ModuleList([models.KPlanesFeaturePlane(32, r) for _ in range(3)]) for r in res])
    x = (torch.rand(256, 3) * 2 - 1)
    x[:8] = torch.tensor([-1., 1., 0.])[torch.randint(0, 3, (8, 3))]
    feat = field(x)
    gup = torch.randn_like(feat)
    feat.backward(gup)
    planes = {f"plane_{s}_{p}": field.planes[s][p].plane for s in range(3) for p in range(3)}
    gplanes = {f"grad_plane_{s}_{p}": field.planes[s][p].plane.grad for s in range(3) for p in range(3)}
    save("G7b_kplanes_field", x=x, feat=feat, grad_feat=gup, loss_tv=field.loss_tv(), loss_l1=field.loss_l1(), **planes, **gplanes)

    # ---- G8: Vanilla feature / sigma / rgb heads (models.py:7-89) ---------------------------
    torch.manual_seed(8)
    fm = models.VanillaFeatureMLP(6, 64, 3)
    od = models.VanillaOpacityDecoder(64)
    cd = models.VanillaColorDecoder(8, 64, 64, 3)
    x = torch.rand(256, 3) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(256, 3), dim=-1)
    feat = fm(x)
    sig = od(feat)
    rgb = cd(feat, dirs)
    gs, gc = torch.randn_like(sig), torch.randn_like(rgb)
    (sig * gs).sum().backward(retain_graph=True)
    g_sigma_params = {"gsig." + n: p.grad.clone() for n, p in list(fm.named_parameters(prefix="fm")) + list(od.named_parameters(prefix="od"))}
    fm.zero_grad(); od.zero_grad()
    (rgb * gc).sum().backward()
    g_rgb_params = {"grgb." + n: p.grad.clone() for n, p in list(fm.named_parameters(prefix="fm")) + list(cd.named_parameters(prefix="cd"))}
    save("G8_vanilla_heads", x=x, dirs=dirs, feat=feat, sigma=sig, rgb=rgb, grad_sigma=gs, grad_rgb=gc,
         **{"fm." + k: v for k, v in sd_np(fm).items()}, **{"od." + k: v for k, v in sd_np(od).items()},
         **{"cd." + k: v for k, v in sd_np(cd).items()}, **g_sigma_params, **g_rgb_params)
    # head fed with free-standing features (decoder-only parity, incl. grads w.r.t. inputs)
    torch.manual_seed(88)
    featin = torch.rand(200, 96, requires_grad=True)
    od2 = models.VanillaOpacityDecoder(96); cd2 = models.VanillaColorDecoder(8, 96, 64, 3)
    d2 = torch.nn.functional.normalize(torch.randn(200, 3), dim=-1)
    s2 = od2(featin); c2 = cd2(featin, d2)
    gs2, gc2 = torch.randn_like(s2), torch.randn_like(c2)
    ((s2 * gs2).sum() + (c2 * gc2).sum()).backward()
    save("G8b_decoders_96", feat=featin, dirs=d2, sigma=s2, rgb=c2, grad_sigma=gs2, grad_rgb=gc2, grad_feat=featin.grad,
         **{"od." + k: v for k, v in sd_np(od2).items()}, **{"cd." + k: v for k, v in sd_np(cd2).items()},
         **{"god." + n: p.grad for n, p in od2.named_parameters()}, **{"gcd." + n: p.grad for n, p in cd2.named_parameters()})

    # ---- G9: NerfRenderer end-to-end (core.py:209-267) with restated weights ----------------
    torch.manual_seed(9)
    field = models.KPlanesFeatureField(32)
    field.planes = torch.nn.ModuleList([torch.nn.ModuleList([models.KPlanesFeaturePlane(32, r) for _ in range(3)]) for r in res])
    od = models.VanillaOpacityDecoder(96); cd = models.VanillaColorDecoder(8, 96, 64, 3)
    with torch.no_grad():
        od.net.net[2].bias.add_(4.0)      # dense enough that early termination triggers
    bg = torch.tensor([1.0, 1.0, 1.0])
    rend = core.NerfRenderer(field, od, cd, bg)
    g = core.OccupancyGrid(32, 1 / 1024.)
    g.grid[:, :, 24:] = 0.
    g.mean = g.grid.mean().item()
    R, S = 48, 64
    o = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1) * 4.0311
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(R, 3), dim=-1)
    m = core.RayMarcherAABB(aabb, S, 0.1)
    rp = core.RayProvider(g, core.ContractionAABB(aabb), m)
    packed, info = rp(o, d, training=False)
    feat = field(packed[:, :3]); sig = od(feat).ravel()
    w = core.NerfWeights.apply(sig, packed[:, 6], info, 1e-4)
    out = rend(packed, info)
    target = torch.rand(R, 3)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    grads = {"grad." + n: p.grad for n, p in rend.named_parameters()}
    out_nobg = core.NerfRenderer(field, od, cd, None)(packed, info)
    save("G9_renderer_kplanes", packed=packed, info=info, bg=bg, sigma=sig, weights=w, rendered=out, rendered_nobg=out_nobg,
         target=target, loss=loss, n_terminated=int((w == 0).sum()),
         **{"sd." + k: v for k, v in sd_np(rend).items()}, **grads)
    # empty-iteration branch (core.py:235-254): N == 0 and all-masked
    empty = rend(torch.zeros(0, 7), torch.zeros(R, 2, dtype=torch.int32))
    save("G9b_renderer_empty", rendered_empty=empty, bg=bg)

    # ---- G11: Cobafa (models.py:209-266) eval mode -----------------------------------------
    torch.manual_seed(11)
    cf = models.CobafaFeatureField(basis_res=[4, 5, 6], coef_res=4, freqs=[2.0, 3.5, 8.0], channels=[2, 2, 2], mlp_hidden_dim=16)
    cf.eval()
    x = torch.rand(128, 3) * 2 - 1
    save("G11_cobafa", x=x, feat=cf(x), freqs=[2.0, 3.5, 8.0], **{"sd." + k: v for k, v in sd_np(cf).items()})

    # ---- G12: ray generation (data.py:48-73) on the reference's own fixture cameras --------
    nd = data.parse_nerf_synthetic(__import__("pathlib").Path(os.path.join(args.ref, "tests", "dummy", "hotdog")), "train")
    ro, rd = nd.generate_rays()
    K = nd.intrinsics
    save("G12_rays_fixture", cameras=nd.cameras, fx=K.fx, fy=K.fy, cx=K.cx, cy=K.cy, w=K.w, h=K.h,
         rays_o_0=ro[0][::25, ::25], rays_d_0=rd[0][::25, ::25], rays_d_1=rd[1][::25, ::25], scene_scale=nd.scene_scale(),
         img0_sub=nd.imgs[0][::25, ::25], bg=nd.bg_color)


    # ---- G13: explicit K-Planes decoders (models.py:183-205; exercised by tests/test_models.py:35-69) -------------
    torch.manual_seed(13)
    featin = torch.rand(200, 96, requires_grad=True)
    eo = models.KPlanesExplicitOpacityDecoder(96); ec = models.KPlanesExplicitColorDecoder(96, 8, 128)
    d3 = torch.nn.functional.normalize(torch.randn(200, 3), dim=-1)
    with torch.no_grad():
        eo.net.weight.mul_(0.1)                      # x = <f, W f + b> stays inside exp's comfortable range
    s3 = eo(featin); c3 = ec(featin, d3)
    gs3, gc3 = torch.randn_like(s3), torch.randn_like(c3)
    ((s3 * gs3).sum() + (c3 * gc3).sum()).backward()
    save("G13_explicit_decoders", feat=featin, dirs=d3, sigma=s3, rgb=c3, grad_sigma=gs3, grad_rgb=gc3, grad_feat=featin.grad,
         **{"eo." + k: v for k, v in sd_np(eo).items()}, **{"ec." + k: v for k, v in sd_np(ec).items()},
         **{"geo." + n: p.grad for n, p in eo.named_parameters()}, **{"gec." + n: p.grad for n, p in ec.named_parameters()})

    # ---- G14: NerfRenderer over the Vanilla stack of run.py:131-134 (VanillaFeatureMLP(10, 256, 8)) -------------------
    torch.manual_seed(14)
    fm = models.VanillaFeatureMLP(10, 256, 8)
    od = models.VanillaOpacityDecoder(256); cd = models.VanillaColorDecoder(8, 256, 64, 3)
    with torch.no_grad():
        # dense enough that about half of the samples sit behind a terminated ray (w == 0 tails -> the boolean gather of
        # core.py:246-249), thin enough that the fp32 suffix-sum cancellation of cuda.cu:49-56 stays below 1e-4 of the gradients
        # (at sigma = 30 the reference's own fp32 backward is 1e-3 away from an fp64 evaluation of the same formula)
        for lin in [m_ for m_ in fm.net.net.modules() if isinstance(m_, torch.nn.Linear)][1:]:
            lin.weight.mul_(2.4)          # default init collapses the variation over x (y would be constant to 1e-4) ...
        od.net.net[0].weight.mul_(8.)     # ... with these gains sigma spreads over [2.8, 8.9]
        od.net.net[2].bias.add_(3.0)
    rendv = core.NerfRenderer(fm, od, cd, bg)
    R, S = 40, 48
    o = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1) * 4.0311
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(R, 3), dim=-1)
    g = core.OccupancyGrid(32, 1 / 1024.)
    g.grid[:, :, 22:] = 0.
    g.mean = g.grid.mean().item()
    rp = core.RayProvider(g, core.ContractionAABB(aabb), core.RayMarcherAABB(aabb, S, 0.1))
    packed, info = rp(o, d, training=False)
    sig = od(fm(packed[:, :3])).ravel()
    w = core.NerfWeights.apply(sig, packed[:, 6], info, 1e-4)
    out = rendv(packed, info)
    target = torch.rand(R, 3)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    save("G14_renderer_vanilla", packed=packed, info=info, bg=bg, weights=w, rendered=out, target=target, loss=loss,
         n_masked=int((w == 0).sum()), **{"sd." + k: v for k, v in sd_np(rendv).items()},
         **{"grad." + n: p.grad for n, p in rendv.named_parameters()})

    # ---- G15: BASELINE config 5 composed: Cobafa field + RayMarcherUnbounded + ContractionMip360(inf) + renderer --------
    # (run.py:141-147,154-156; small grids, eval-mode dropout so that the forward is a function)
    torch.manual_seed(15)
    freqs15 = [2.0, 3.5, 8.0]
    cf = models.CobafaFeatureField(basis_res=[8, 10, 12], coef_res=8, freqs=freqs15, channels=[8, 8, 4], mlp_hidden_dim=128)
    od = models.VanillaOpacityDecoder(128); cd = models.VanillaColorDecoder(8, 128, 64, 3)
    with torch.no_grad():
        od.net.net[2].bias.add_(3.0)
    rendc = core.NerfRenderer(cf, od, cd, None)
    rendc.eval()
    R, S = 48, 40
    o = torch.rand(R, 3) * 0.6 - 0.3                                   # cameras inside the scene (unbounded capture)
    d = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    g = core.OccupancyGrid(24, 1.3 / S)
    kk = torch.randint(0, 30, (24, 24, 24))
    g.grid.copy_(torch.tensor(decay) ** kk.float())
    g.mean = g.grid.mean().item()
    mu = core.RayMarcherUnbounded(S, 0.1, 1e5, 1.3)
    rpu = core.RayProvider(g, core.ContractionMip360(float("inf")), mu)
    packed, info = rpu(o, d, training=False)
    out = rendc(packed, info)
    target = torch.rand(R, 3)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    save("G15_config5_cobafa_unbounded", rays_o=o, rays_d=d, n_samples=S, near=0.1, uniform_range=1.3, grid=g.grid, threshold=g.threshold,
         packed=packed, info=info, rendered=out, target=target, loss=loss, freqs=freqs15,
         **{"sd." + k: v for k, v in sd_np(rendc).items()}, **{"grad." + n: p.grad for n, p in rendc.named_parameters()})

    # ---- G16: config 5 again with a MODERATE medium: some rays terminate (w == 0 tails, core.py:243-249) but the fp32 suffix
    # sums of cuda.cu:49-56 stay well conditioned (< 1e-4 of every gradient tensor), so that the Cobafa sigma path -- coefficient
    # and basis grid gradients through terminated rays -- is pinned as tightly as every other row (G15's dense medium lets the
    # reference's own backward drift by 4e-2 on the sigma head).  Own seed: the sections above are unchanged.
    torch.manual_seed(16)
    freqs16 = [2.0, 3.5, 8.0]
    cf = models.CobafaFeatureField(basis_res=[8, 10, 12], coef_res=8, freqs=freqs16, channels=[8, 8, 4], mlp_hidden_dim=128)
    od = models.VanillaOpacityDecoder(128); cd = models.VanillaColorDecoder(8, 128, 64, 3)
    with torch.no_grad():
        od.net.net[2].bias.add_(3.0)
    rendm = core.NerfRenderer(cf, od, cd, None)
    rendm.eval()
    R, S = 48, 40
    o = torch.rand(R, 3) * 0.6 - 0.3
    d = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    g = core.OccupancyGrid(24, 1.3 / S)
    kk = torch.randint(0, 30, (24, 24, 24))
    g.grid.copy_(torch.tensor(decay) ** kk.float())
    cut = 5      # far samples (large steps of the unbounded marcher) are culled by the grid
    keep = torch.zeros(24, 24, 24, dtype=torch.bool)
    keep[cut:24 - cut, cut:24 - cut, cut:24 - cut] = True
    g.grid.mul_(keep.float())
    g.mean = g.grid.mean().item()
    mu = core.RayMarcherUnbounded(S, 0.1, 1e5, 1.3)
    rpm = core.RayProvider(g, core.ContractionMip360(float("inf")), mu)
    packed, info = rpm(o, d, training=False)
    sig = od(cf(packed[:, :3])).ravel()
    w = core.NerfWeights.apply(sig, packed[:, 6], info, 1e-4)
    out = rendm(packed, info)
    target = torch.rand(R, 3)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    ends = info[:, 0] + info[:, 1]
    n_term = int(sum(1 for r_ in range(R) if info[r_, 1] > 0 and w[ends[r_] - 1] == 0))
    save("G16_config5_moderate", rays_o=o, rays_d=d, n_samples=S, near=0.1, uniform_range=1.3,
         grid=g.grid, threshold=g.threshold, packed=packed, info=info, weights=w, n_masked=int((w == 0).sum()), n_terminated_rays=n_term,
         rendered=out, target=target, loss=loss, freqs=freqs16,
         **{"sd." + k: v for k, v in sd_np(rendm).items()}, **{"grad." + n: p.grad for n, p in rendm.named_parameters()})

    with open(os.path.join(out_dir, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by oracle/make_goldens.py from the reference imported at %s\n" % args.ref)
        f.write("torch %s, numpy %s, 1 CPU thread, _cuda replaced by oracle/weights_ref.c\n" % (torch.__version__, np.__version__))
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
