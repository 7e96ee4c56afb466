"""GPU parity, round-2 additions (all through the C ABI):

* G13: explicit K-Planes decoders (reference models.py:183-205, exercised by its tests/test_models.py:35-69);
* truncated exponential outside its clamp (models.py:42-55, |x| > 15);
* G14: NerfRenderer over VanillaFeatureMLP(10, 256, 8) with most samples masked (core.py:243-249);
* G15: BASELINE config 5 composed -- unbounded marcher + inf-norm Mip-NeRF-360 contraction + Cobafa field + renderer;
* K-Planes backward at the full 128/256/512 resolution on a ray-ordered slice of a real dynamic batch against ATen's CPU
  grid_sampler_2d backward (what the reference runs);
* the device-side dynamic-batch rule (tn_batch_plan) against the oracle's restatement of run.py:215-244 on random counts;
* the ctypes stub of INTEGRATION.md section 1, executed verbatim.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from _ties import assert_grads_match_up_to_relu_ties
from conftest import load_golden
from oracle import tinynerf_oracle as orc
from oracle import torch_port as tp

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(DEV)


def sub(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(v) for k, v in g.items() if k.startswith(prefix)}


def close_rel_inf(got, ref, rel, name=""):
    """|got - ref| <= rel * max|ref| element-wise: the bound for quantities that are sums over samples, where the summation
    order (atomics, MFMA K-order) moves every element by a few ulp OF THE LARGEST terms, not of the element itself."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=0, atol=rel * max(float(np.abs(ref).max()), 1e-30), err_msg=name)


# ------------------------------------------------------------------------------------------------ G13
def test_explicit_decoders_vs_reference():
    from tinynerf_amd import models as m
    g = load_golden("G13_explicit_decoders")
    eo, ec = m.KPlanesExplicitOpacityDecoder(96), m.KPlanesExplicitColorDecoder(96, 8, 128)
    eo.load_state_dict(sub(g, "eo.")); ec.load_state_dict(sub(g, "ec."))        # reference checkpoint keys load unchanged
    eo.to(DEV); ec.to(DEV)
    feat = cu(g["feat"]).requires_grad_(True)
    s, c = eo(feat), ec(feat, cu(g["dirs"]))
    assert s.shape == (200, 1) and c.shape == (200, 3)                           # reference tests/test_models.py:35-69
    np.testing.assert_allclose(s.detach().cpu().numpy(), g["sigma"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["rgb"], rtol=0, atol=TOL)
    ((s * cu(g["grad_sigma"])).sum() + (c * cu(g["grad_rgb"])).sum()).backward()
    got = {"feat": feat.grad.cpu().numpy(), **{"eo." + k: p.grad.cpu().numpy() for k, p in eo.named_parameters()},
           **{"ec." + k: p.grad.cpu().numpy() for k, p in ec.named_parameters()}}
    for k, v in got.items():                                                     # the golden pins the port (tests/test_oracle_golden_r2.py)
        assert v.shape == (g["grad_feat"] if k == "feat" else g[("geo." if k[:2] == "eo" else "gec.") + k[3:]]).shape

    def ref():
        leaves = {"eo." + k: v.clone().requires_grad_(True) for k, v in sub(g, "eo.").items()}
        leaves.update({"ec." + k: (v.clone().requires_grad_(True) if not k.endswith("freqs") else v) for k, v in sub(g, "ec.").items()})
        f = torch.as_tensor(g["feat"]).requires_grad_(True)
        so, co = tp.explicit_sigma(leaves, f, "eo."), tp.explicit_rgb(leaves, f, torch.as_tensor(g["dirs"]), "ec.")
        ((so * torch.as_tensor(g["grad_sigma"])).sum() + (co * torch.as_tensor(g["grad_rgb"])).sum()).backward()
        return {"feat": f.grad.numpy(), **{k: v.grad.numpy() for k, v in leaves.items() if v.requires_grad}}
    # 2e-5 of each tensor's largest element: sums over 200 samples in rocBLAS / MFMA order vs ATen's; with no tie unit flipped
    # the HIP gradients are held against the golden's own values (geo.* / gec.* / grad_feat), not only against the port
    golden = {k: (g["grad_feat"] if k == "feat" else g[("geo." if k[:2] == "eo" else "gec.") + k[3:]]) for k in got}
    assert_grads_match_up_to_relu_ties(got, ref, 2e-5, golden=golden)


def test_explicit_decoders_reference_shape_tests():
    """the reference's own tests/test_models.py:35-69 (test_kplanes, test_kplanes_hybrid) on the HIP modules"""
    from tinynerf_amd import models as m
    field = m.KPlanesFeatureField(32).to(DEV)
    od = m.KPlanesExplicitOpacityDecoder(feature_dim=field.feature_dim).to(DEV)
    n_rays = 100
    rays_o, rays_d = torch.rand(n_rays, 3, device=DEV), torch.rand(n_rays, 3, device=DEV)
    features = field(rays_o)
    opacity = od(features)
    assert features.size() == (n_rays, field.feature_dim) and opacity.size() == (n_rays, 1)
    for cd in (m.KPlanesExplicitColorDecoder(field.feature_dim, 4, 128).to(DEV), m.VanillaColorDecoder(4, field.feature_dim, 128, 3).to(DEV)):
        assert cd(features, rays_d).size() == (n_rays, 3)
    assert field.loss_l1().item() >= 0. and field.loss_tv().item() >= 0.


def test_truncated_exponential_clamp_edges():
    """models.py:42-55: forward exp(x), backward g * exp(clamp(x, -15, 15)); |x| > 15 is where the two differ."""
    from tinynerf_amd import models as m
    x = torch.tensor([-40.0, -15.5, -15.0, -3.0, 0.0, 2.5, 15.0, 15.5, 20.0, 30.0], device=DEV, requires_grad=True)
    y = m.truncated_exp(x)
    g = torch.linspace(0.5, 2.0, x.numel(), device=DEV)
    y.backward(g)
    xr = x.detach().cpu().double()
    np.testing.assert_allclose(y.detach().cpu().numpy(), torch.exp(xr).numpy(), rtol=2e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), (g.cpu().double() * torch.exp(xr.clamp(-15, 15))).numpy(), rtol=2e-6)
    # x reaches expf() unmodified (no x + 1 - 1 round trip, which loses ulp(x + 1) / 2 of x where x + 1 crosses a binade).  What
    # is left is the device expf itself: exp2(x log2 e) with the product rounded in fp32, i.e. a relative error of up to
    # ~6e-8 |x| (measured: 6.8e-7 at x = 80, 1.5e-6 at x = 64) -- the bound below
    x2 = torch.tensor([63.9999962, 31.9999981, -61.987654, 80.25, 37.123456, 5.4321e-5, 15.9999990, 1e-9, -3e-8], device=DEV)
    y2 = m.truncated_exp(x2).cpu().double().numpy()
    ref2 = torch.exp(x2.cpu().double()).numpy()
    assert np.all(np.abs(y2 - ref2) <= (3e-7 + 6e-8 * np.abs(x2.cpu().numpy())) * ref2), (y2, ref2)
    # the same clamp inside the fused sigma head (TN_ACT_EXP_M1 backward): pre-activations pushed beyond +-15 by the bias
    torch.manual_seed(0)
    od = m.VanillaOpacityDecoder(32).to(DEV)
    feat = torch.rand(64, 32, device=DEV)
    for shift in (25.0, -25.0):
        with torch.no_grad():
            od.net.net[2].bias.fill_(shift)
        od.zero_grad()
        s = od(feat)
        s.backward(torch.ones_like(s))
        sd = {k: v.detach().cpu() for k, v in od.state_dict().items()}
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = tp._TruncExp.apply(tp.mlp(leaves, "net.net.", feat.cpu()) - 1.)
        ref.backward(torch.ones_like(ref))
        np.testing.assert_allclose(s.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5)
        for k, p in od.named_parameters():
            r = leaves[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-5 * np.abs(r).max(), err_msg=f"{k} shift {shift}")


# ------------------------------------------------------------------------------------------------ G14
def _vanilla_renderer(g):
    from tinynerf_amd import core, models as m
    r = core.NerfRenderer(m.VanillaFeatureMLP(10, 256, 8), m.VanillaOpacityDecoder(256), m.VanillaColorDecoder(8, 256, 64, 3), cu(g["bg"]))
    r.load_state_dict(sub(g, "sd."))
    return r.to(DEV)


@pytest.mark.parametrize("fused", [True, False])
def test_vanilla_renderer_vs_reference(fused, matmul):
    """fused: heads + scan + composite as one autograd node, every sample through the colour head (tinynerf_amd.fused);
    not fused: module by module with the boolean gather of core.py:246-249."""
    from tinynerf_amd import core
    g = load_golden("G14_renderer_vanilla")
    r = _vanilla_renderer(g)
    r.fused = fused
    packed, info = cu(g["packed"]), cu(g["info"], torch.int32)
    with torch.no_grad():
        sig = r.sigma_decoder(r.feature_module(packed[:, :3])).ravel()
        w = core.NerfWeights.apply(sig, packed[:, 6].contiguous(), info, 1e-4)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], rtol=0, atol=TOL)
    assert abs(int((w == 0).sum()) - int(g["n_masked"])) <= 2 and int(g["n_masked"]) > 300    # the colour head runs on ~60 %
    out = r(packed, info)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["rendered"], rtol=0, atol=TOL)
    loss = torch.nn.functional.mse_loss(out, cu(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    got = {name: p.grad.cpu().numpy() for name, p in r.named_parameters()}
    sd = sub(g, "sd.")
    pk, inf_, bg, target = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"]), torch.as_tensor(g["bg"]), torch.as_tensor(g["target"])

    def ref():                                                                    # the CPU port, pinned to G14 by tests/test_oracle_golden_r2.py
        return tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pk, inf_, bg, vanilla_freqs=10), target))[0]
    # ten 256-wide layers deep, every one with its own fp32 summation order: 1e-4 of each tensor's largest element
    assert_grads_match_up_to_relu_ties(got, ref, 1e-4, weights_conditioning=True, golden={n: g["grad." + n] for n in got}, cond_cap=1e-4)


# ------------------------------------------------------------------------------------------------ G15 (BASELINE config 5)
@pytest.mark.parametrize("fused", [True, False])
def test_config5_sampler_and_renderer_vs_reference(fused, matmul):
    from tinynerf_amd import core, models as m
    g = load_golden("G15_config5_cobafa_unbounded")
    S = int(g["n_samples"])
    grid = core.OccupancyGrid(24, float(g["uniform_range"]) / S).to(DEV)
    grid.grid.copy_(cu(g["grid"]))
    grid.mean = float(grid.grid.mean().item())
    assert grid.threshold == pytest.approx(float(g["threshold"]))
    marcher = core.RayMarcherUnbounded(S, float(g["near"]), 1e5, float(g["uniform_range"]))
    prov = core.RayProvider(grid, core.ContractionMip360(float("inf")), marcher)
    packed, info = prov(cu(g["rays_o"]), cu(g["rays_d"]), training=False)
    assert np.array_equal(info.cpu().numpy(), g["info"])                          # bit-exact ints
    assert np.array_equal(packed[:, :6].cpu().numpy().view(np.int32), g["packed"][:, :6].view(np.int32))
    np.testing.assert_allclose(packed[:, 6].cpu().numpy(), g["packed"][:, 6], rtol=2e-3)      # device vs host linspace
    freqs = [float(f) for f in g["freqs"]]
    cf = m.CobafaFeatureField(basis_res=[8, 10, 12], coef_res=8, freqs=freqs, channels=[8, 8, 4], mlp_hidden_dim=128)
    r = core.NerfRenderer(cf, m.VanillaOpacityDecoder(128), m.VanillaColorDecoder(8, 128, 64, 3), None)
    r.load_state_dict(sub(g, "sd."))
    r.to(DEV).eval()                                                                   # Dropout(0.01) off, as in the golden
    r.fused = fused
    pk, inf_ = cu(g["packed"]), cu(g["info"], torch.int32)
    out = r(pk, inf_)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["rendered"], rtol=0, atol=TOL)
    loss = torch.nn.functional.mse_loss(out, cu(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    got = {name: p.grad.cpu().numpy() for name, p in r.named_parameters()}
    sd = sub(g, "sd.")
    pc, ic, target = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"]), torch.as_tensor(g["target"])

    def ref():                                                                    # the CPU port, pinned to G15 by tests/test_oracle_golden_r2.py
        return tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pc, ic, None, cobafa_freqs=freqs), target))[0]
    # Half of this fixture's samples sit behind a terminated ray, with steps up to 13.8 (unbounded marcher): there the
    # reference's fp32 weights backward (cuda.cu:49-56, acc = -sum + prefix) is rounding noise times the step size, and its
    # gradients are 4e-2 (sigma head) / 1e-3 (grids) away from an exact evaluation of its own formula.  The tolerance per
    # tensor is therefore max(2e-5, 4 x that distance): colour head strict, sigma path as loose as the reference itself is.
    cond = tp.weights_conditioning(ref)
    assert cond["rgb_decoder.net.net.0.weight"] == 0.0 and cond["sigma_decoder.net.net.2.bias"] > 1e-3
    # Round-3 verdict: a 16 % tolerance pins nothing.  On THIS medium only the tensors the reference itself determines to better
    # than the cap are compared (the colour head at 2e-5: it does not pass through the weights backward; whatever else is
    # conditioned below 2e-3: the coefficient grid and most of the 128-wide stack); the sigma path of config 5's composition is pinned by the thin-medium evaluation below
    # (every tensor, < 4e-4) and by G16 (terminated rays, cap 6e-3) -- not by a loose assert here.
    CAP = 8e-3
    pinned = sorted(k for k, c in cond.items() if 4.0 * c <= CAP)
    assert any(k.startswith("rgb_decoder") for k in pinned) and len(pinned) >= 10, pinned

    def only(dct):
        return {k: dct[k] for k in pinned}
    assert_grads_match_up_to_relu_ties(only(got), lambda: only(ref()), {k: max(2e-5, 4.0 * cond[k]) for k in pinned},
                                       golden={n: g["grad." + n] for n in pinned}, cond_cap=CAP)
    # (tightly pinned twin of this fixture with terminated rays: G16, test_config5_moderate_medium_vs_reference)
    # the same composition where the reference is well conditioned: thin medium (sigma bias - 4: no ray terminates),
    # every tensor to 2e-5 ... 4 x its (small) conditioning against the CPU port
    sd2 = dict(sd)
    sd2["sigma_decoder.net.net.2.bias"] = sd["sigma_decoder.net.net.2.bias"] - 4.0
    r.load_state_dict(sd2)
    r.zero_grad()
    torch.nn.functional.mse_loss(r(pk, inf_), cu(g["target"])).backward()
    got2 = {name: p.grad.cpu().numpy() for name, p in r.named_parameters()}

    def ref2():
        return tp.grads_of(sd2, lambda p: torch.nn.functional.mse_loss(tp.render(p, pc, ic, None, cobafa_freqs=freqs), target))[0]
    cond2 = tp.weights_conditioning(ref2)
    assert max(cond2.values()) < 1e-4
    assert_grads_match_up_to_relu_ties(got2, ref2, {k: max(2e-5, 4.0 * c) for k, c in cond2.items()})


@pytest.mark.parametrize("fused", [True, False])
def test_config5_moderate_medium_vs_reference(fused, matmul):
    """G16: config 5's composition (unbounded marcher + inf-norm Mip-360 contraction + Cobafa + renderer) with a medium in which
    33 of 48 rays terminate (127 samples with w == 0) while the reference's fp32 weights backward stays well conditioned: the
    far samples, whose steps of up to 13.8 amplify the suffix-sum cancellation of cuda.cu:49-56 in G15, are culled by the grid.
    Cobafa's coefficient / basis grids and its 128-wide stack -- the sigma path through terminated rays -- are pinned to
    max(2e-5, 4 x 6e-5) of each tensor, the sigma head to 4 x 1.1e-3 (G15: 16 %)."""
    from tinynerf_amd import core, models as m
    g = load_golden("G16_config5_moderate")
    S = int(g["n_samples"])
    grid = core.OccupancyGrid(24, float(g["uniform_range"]) / S).to(DEV)
    grid.grid.copy_(cu(g["grid"]))
    grid.mean = float(grid.grid.mean().item())
    prov = core.RayProvider(grid, core.ContractionMip360(float("inf")), core.RayMarcherUnbounded(S, float(g["near"]), 1e5, float(g["uniform_range"])))
    packed, info = prov(cu(g["rays_o"]), cu(g["rays_d"]), training=False)
    assert np.array_equal(info.cpu().numpy(), g["info"])                          # bit-exact ints on the culled grid
    assert np.array_equal(packed[:, :6].cpu().numpy().view(np.int32), g["packed"][:, :6].view(np.int32))
    freqs = [float(f) for f in g["freqs"]]
    cf = m.CobafaFeatureField(basis_res=[8, 10, 12], coef_res=8, freqs=freqs, channels=[8, 8, 4], mlp_hidden_dim=128)
    r = core.NerfRenderer(cf, m.VanillaOpacityDecoder(128), m.VanillaColorDecoder(8, 128, 64, 3), None)
    r.load_state_dict(sub(g, "sd."))
    r.to(DEV).eval()
    r.fused = fused
    pk, inf_ = cu(g["packed"]), cu(g["info"], torch.int32)
    with torch.no_grad():
        sig = r.sigma_decoder(r.feature_module(pk[:, :3])).ravel()
        w = core.NerfWeights.apply(sig, pk[:, 6].contiguous(), inf_, 1e-4)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], rtol=0, atol=TOL)
    assert int(g["n_terminated_rays"]) >= 30 and abs(int((w == 0).sum()) - int(g["n_masked"])) <= 2
    out = r(pk, inf_)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["rendered"], rtol=0, atol=TOL)
    loss = torch.nn.functional.mse_loss(out, cu(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    got = {name: p.grad.cpu().numpy() for name, p in r.named_parameters()}
    sd = sub(g, "sd.")
    pc, ic, target = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"]), torch.as_tensor(g["target"])

    def ref():                                                                    # the CPU port, pinned to G16 by tests/test_oracle_golden_r2.py
        return tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pc, ic, None, cobafa_freqs=freqs), target))[0]
    cond = tp.weights_conditioning(ref)
    assert max(c for k, c in cond.items() if k.startswith("feature_module")) < 1e-4       # the Cobafa sigma path: well conditioned
    assert_grads_match_up_to_relu_ties(got, ref, {k: max(2e-5, 4.0 * c) for k, c in cond.items()}, golden={n: g["grad." + n] for n in got},
                                       cond_cap=6e-3)


def test_config5_training_matches_cpu_port():
    """Cobafa + scene_type="unbounded" (run.py:141-147,154-156) through Trainer.step against the CPU port of train()."""
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=2, res=48, seed=9, device="cpu")
    o = (o * 0.08).contiguous()                                                      # cameras inside the scene (unbounded capture)
    n_steps = 8
    cfg = TrainConfig(method="cobafa", scene_type="unbounded", batch_size=256, n_samples=32, seed=5, occupancy_res=32,
                      deterministic=True, scene_scale=1.3)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), None, torch.device(DEV))
    tr.renderer.feature_module.dropout.p = 0.0                                      # the process RNG cannot be shared with the port
    sd0 = {k: v.detach().cpu().contiguous().clone() for k, v in tr.renderer.state_dict().items()}
    ref_losses, ref_sd, ref_counts = tp.reference_training(
        sd0, o.numpy(), d.numpy(), rgb.numpy(), method="cobafa", batch_size=256, n_samples=32, n_steps=n_steps, occupancy_res=32,
        bg=None, scene_type="unbounded", scene_scale=1.3, cobafa_freqs=tr.renderer.feature_module.freqs)
    losses, counts = [], []
    for _ in range(n_steps):
        st = tr.step()
        losses.append(tr.loss_value())
        counts.append((int(st["n_samples"]), int(st["n_rays"])))
    assert counts[0] == ref_counts[0]
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(losses, ref_losses, rtol=3e-2)
    assert counts[:3] == ref_counts[:3], (counts, ref_counts)


# ------------------------------------------------------------------------------------------------ K-Planes backward, full resolution
def test_kplanes_backward_full_resolution_vs_grid_sampler():
    """The run-merging / neighbour-carry scatter of kplanes.hip on the planes the reference trains (128/256/512, 32 channels)
    with ray-ordered samples of a real dynamic batch -- consecutive samples of a ray fall into the same or adjacent texels,
    which is exactly what the merging exploits -- against torch CPU autograd of 9 x grid_sample (models.py:105-113,153-163)."""
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer
    dev = torch.device(DEV)
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=2, res=200, seed=1, device=DEV)
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=512, seed=0, deterministic=True)
    tr = Trainer(cfg, o, d, rgb, torch.ones(3, device=dev), dev)
    lin = torch.linspace(-1, 1, 128, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    tr.occupancy_grid.grid.copy_(torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, tr.occupancy_grid.decay ** 20))
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    tr._cursor = 200 * 90                                                           # rows through the middle of the image
    packed, info, target, k = tr.build_batch()
    n = min(packed.size(0), 24000)
    assert n >= 20000, n
    x = packed[:n, :3].clone()
    field = tr.renderer.feature_module
    planes = field.plane_tensors()
    torch.manual_seed(2)
    gfeat = torch.randn(n, 96, device=dev)
    for p in planes:
        p.grad = None
    feat = field(x)
    feat.backward(gfeat)
    sd = {f"feature_module.planes.{s}.{p}.plane": planes[3 * s + p].detach().cpu().contiguous().clone().requires_grad_(True)
          for s in range(3) for p in range(3)}
    ref = tp.kplanes_features(sd, x.cpu())
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=TOL)
    ref.backward(gfeat.cpu())
    for s in range(3):
        for p in range(3):
            r = sd[f"feature_module.planes.{s}.{p}.plane"].grad.numpy()
            got = planes[3 * s + p].grad.cpu().numpy()
            assert np.count_nonzero(r) > 1000
            close_rel_inf(got, r, 1e-5, f"plane {s}.{p}")
            assert np.array_equal(got != 0, r != 0) or np.abs(got[(got != 0) != (r != 0)]).max() < 1e-5 * np.abs(r).max()


# ------------------------------------------------------------------------------------------------ a8: tn_batch_plan
@pytest.mark.parametrize("seed", range(6))
def test_batch_plan_vs_oracle_random(seed):
    """run.py:215-244 on the device against the oracle's loop, random per-ray counts: sparse / dense / empty batches, rule
    tripping on the first batch, on the last one, and not at all."""
    from tinynerf_amd import _lib as L
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 7, 64, 256]))
    n_b = int(rng.integers(1, 40))
    S = int(rng.choice([8, 64, 300]))
    dens = rng.random(n_b)[:, None] * (rng.random((n_b, B)) < rng.random()) if seed % 2 else rng.random((n_b, B))
    counts = np.floor(dens * S).astype(np.int32)
    if seed == 3:
        counts[:2] = 0                                                              # empty leading batches
    target = int(rng.integers(1, max(2, int(counts.sum() * 1.3) + 2)))
    if seed == 5:
        target = int(counts.sum()) * 4 + 10                                         # never trips

    def batches():
        for b in range(n_b):
            yield b, None, np.zeros((B, 3), np.float32)

    def provider(b, _):
        c = counts[b]
        return np.zeros((int(c.sum()), 7), np.float32), np.stack([np.cumsum(c) - c, c], -1).astype(np.int32)

    try:
        _, info_ref, _, k_ref = orc.dynamic_batch(batches(), provider, target)
        tripped_ref, n_ref = 1, int(info_ref[:, 1].sum())
    except StopIteration:                                                            # the loader ran dry before the rule tripped
        k_ref, tripped_ref, n_ref = n_b, 0, int(counts.sum())
    plan = torch.zeros(4, dtype=torch.int32, device=DEV)
    c_dev = cu(counts.reshape(-1), torch.int32)
    L.call("tn_batch_plan", torch.device(DEV), L.ptr(c_dev), C.c_int64(n_b * B), C.c_int32(B), C.c_int64(target), L.ptr(plan))
    k, n, R, tripped = plan.tolist()
    assert (k, n, R, tripped) == (k_ref, n_ref, k_ref * B, tripped_ref)
    # the training step's form: the rule and the scan of every candidate ray in one multi-workgroup launch
    plan2 = torch.zeros(4, dtype=torch.int32, device=DEV)
    info = torch.full((n_b * B, 2), -1, dtype=torch.int32, device=DEV)
    L.call("tn_batch_plan_scan", torch.device(DEV), L.ptr(c_dev), C.c_int64(n_b * B), C.c_int32(B), C.c_int64(target), L.ptr(plan2), L.ptr(info))
    assert plan2.tolist() == [k, n, R, tripped]
    flat = counts.reshape(-1).astype(np.int64)
    np.testing.assert_array_equal(info.cpu().numpy(), np.stack([np.cumsum(flat) - flat, flat], -1).astype(np.int32))


@pytest.mark.parametrize("n_rays,B", [(1, 1), (4097, 1024), (50_000, 1024), (49_999, 5000), (300 * 64, 64), (260 * 16 + 3, 16)])
def test_batch_plan_scan_shapes(n_rays, B):
    """ragged last batch, batches larger than one 4096-ray sweep, and the > 256-batch fallback to the two single launches"""
    from tinynerf_amd import _lib as L
    rng = np.random.default_rng(n_rays)
    counts = (rng.integers(0, 200, n_rays) * (rng.random(n_rays) < 0.4)).astype(np.int32)
    target = int(counts.sum() * 0.6) + 1
    c_dev = cu(counts, torch.int32)
    plan, plan2 = torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV)
    info = torch.full((n_rays, 2), -1, dtype=torch.int32, device=DEV)
    L.call("tn_batch_plan", torch.device(DEV), L.ptr(c_dev), C.c_int64(n_rays), C.c_int32(B), C.c_int64(target), L.ptr(plan))
    L.call("tn_batch_plan_scan", torch.device(DEV), L.ptr(c_dev), C.c_int64(n_rays), C.c_int32(B), C.c_int64(target), L.ptr(plan2), L.ptr(info))
    assert plan.tolist() == plan2.tolist()
    flat = counts.astype(np.int64)
    np.testing.assert_array_equal(info.cpu().numpy(), np.stack([np.cumsum(flat) - flat, flat], -1).astype(np.int32))


@pytest.mark.parametrize("n_rays", [4096, 4097, 12_345, 640_000, (1 << 21) + 5])
def test_sample_scan_multi_workgroup(n_rays):
    """core.py:179-181 at image size: the multi-workgroup scan (4096 < R <= 2^21, aligned counts), its single-workgroup form on
    either side of that range and on an unaligned view, all against numpy, with and without the base offset of run.py:231"""
    from tinynerf_amd import _lib as L
    rng = np.random.default_rng(n_rays)
    counts = (rng.integers(0, 900, n_rays + 1) * (rng.random(n_rays + 1) < 0.3)).astype(np.int32)
    c_all = cu(counts, torch.int32)
    base = torch.tensor([12345], dtype=torch.int32, device=DEV)
    for off in (0, 1):
        c_dev = c_all[off:off + n_rays]
        ref = counts[off:off + n_rays].astype(np.int64)
        for b in (None, base):
            info = torch.full((n_rays, 2), -1, dtype=torch.int32, device=DEV)
            total = torch.zeros(1, dtype=torch.int32, device=DEV)
            L.call("tn_sample_scan", torch.device(DEV), L.ptr(c_dev), C.c_int64(n_rays), L.ptr(b) if b is not None else C.c_void_p(None), L.ptr(info),
                   L.ptr(total))
            np.testing.assert_array_equal(info.cpu().numpy(), np.stack([np.cumsum(ref) - ref + (12345 if b is not None else 0), ref], -1).astype(np.int32))
            assert int(total.item()) == int(ref.sum())


# ------------------------------------------------------------------------------------------------ INTEGRATION.md section 1
def test_integration_stub_verbatim():
    """The ctypes stub a reference maintainer would paste over src/core.py:7, executed as written in INTEGRATION.md."""
    from tinynerf_amd import _lib as L
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"## 1\..*?```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('path="libtinynerf_hip.so"', f'path="{L.LIB_PATH}"')
    ns = {}
    exec(compile(code, "INTEGRATION.md#1", "exec"), ns)                             # noqa: S102 -- documentation under test
    _cuda = ns["_cuda"]
    rng = np.random.default_rng(0)
    cnt = rng.integers(0, 30, 50).astype(np.int32)
    start = (np.cumsum(cnt) - cnt).astype(np.int32) + 5                              # samples 0..4 and the tail belong to no ray
    n = int(cnt.sum()) + 12
    info = np.stack([start, cnt], -1)
    s = (rng.random(n) * 20).astype(np.float32)
    d = (rng.random(n) * 0.05 + 0.001).astype(np.float32)
    w = _cuda.compute_weights_fwd(cu(s), cu(d), cu(info, torch.int32), 1e-4)
    w_ref = orc.weights_fwd(s, d, info, 1e-4)                                        # zeros where no ray owns the sample (cuda.cu:84)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref, rtol=0, atol=1e-6)
    assert float(w[:5].abs().max()) == 0.0 and float(w[-7:].abs().max()) == 0.0
    g = rng.standard_normal(n).astype(np.float32)
    gs = _cuda.compute_weights_bwd(cu(s), cu(d), cu(info, torch.int32), w, cu(g))
    np.testing.assert_allclose(gs.cpu().numpy(), orc.weights_bwd(s, d, info, w_ref, g), rtol=1e-4, atol=1e-6)
    # no rays at all: zeros like the reference, nothing launched
    w0 = _cuda.compute_weights_fwd(cu(s), cu(d), torch.zeros((0, 2), dtype=torch.int32, device=DEV), 1e-4)
    assert float(w0.abs().max()) == 0.0
