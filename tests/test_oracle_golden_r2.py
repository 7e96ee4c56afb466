"""Pin the CPU port (oracle/torch_port.py) and the sampler oracle to the round-2 goldens captured from the reference:
G13 explicit K-Planes decoders (models.py:183-205), G14 NerfRenderer over VanillaFeatureMLP(10, 256, 8) with 40 % of the samples
masked (core.py:243-249), G15 BASELINE config 5 composed (Cobafa field + RayMarcherUnbounded + ContractionMip360(inf))."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tinynerf_oracle as orc
from oracle import torch_port as tp


def _sub(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(v) for k, v in g.items() if k.startswith(prefix)}


def test_port_explicit_decoders_match_reference():
    g = load_golden("G13_explicit_decoders")
    eo, ec = _sub(g, "eo."), _sub(g, "ec.")
    leaves = {"eo." + k: v.clone().requires_grad_(True) for k, v in eo.items()}
    leaves.update({"ec." + k: (v.clone().requires_grad_(True) if not k.endswith("freqs") else v) for k, v in ec.items()})
    feat = torch.as_tensor(g["feat"]).requires_grad_(True)
    dirs = torch.as_tensor(g["dirs"])
    s = tp.explicit_sigma(leaves, feat, "eo.")
    c = tp.explicit_rgb(leaves, feat, dirs, "ec.")
    np.testing.assert_allclose(s.detach().numpy(), g["sigma"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(c.detach().numpy(), g["rgb"], rtol=0, atol=1e-6)
    ((s * torch.as_tensor(g["grad_sigma"])).sum() + (c * torch.as_tensor(g["grad_rgb"])).sum()).backward()
    np.testing.assert_allclose(feat.grad.numpy(), g["grad_feat"], rtol=1e-5, atol=1e-6)
    for k, v in leaves.items():
        if v.requires_grad:
            ref = g[("geo." if k.startswith("eo.") else "gec.") + k[3:]]
            np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_port_vanilla_renderer_matches_reference():
    g = load_golden("G14_renderer_vanilla")
    sd = _sub(g, "sd.")
    packed, info = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"])
    bg, target = torch.as_tensor(g["bg"]), torch.as_tensor(g["target"])
    assert int(g["n_masked"]) > packed.size(0) // 3            # the boolean gather of core.py:246-249 is exercised
    out = tp.render(sd, packed, info, bg, vanilla_freqs=10)
    np.testing.assert_allclose(out.numpy(), g["rendered"], rtol=0, atol=1e-6)
    grads, loss = tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, packed, info, bg, vanilla_freqs=10), target))
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-6)
    for k, v in grads.items():
        ref = g["grad." + k]
        np.testing.assert_allclose(v, ref, rtol=1e-4, atol=1e-7 * max(1.0, np.abs(ref).max() / 1e-3), err_msg=k)
    assert len(grads) == sum(1 for k in g if k.startswith("grad."))


@pytest.mark.parametrize("name", ["G15_config5_cobafa_unbounded", "G16_config5_moderate"])
def test_config5_sampler_and_port_match_reference(name):
    g = load_golden(name)
    # sampler: unbounded marcher + inf-norm Mip-NeRF-360 contraction + occupancy test, bit-exact ints and coordinates
    packed, info = orc.ray_provider(g["rays_o"], g["rays_d"], marcher="unbounded", contraction="mip360", grid=g["grid"],
                                    threshold=float(g["threshold"]), n_samples=int(g["n_samples"]), near=float(g["near"]),
                                    far=1e5, uniform_range=float(g["uniform_range"]), order=float("inf"))
    assert np.array_equal(info, g["info"])
    assert np.array_equal(packed[:, :6].view(np.int32), g["packed"][:, :6].view(np.int32))
    np.testing.assert_allclose(packed[:, 6], g["packed"][:, 6], rtol=2e-3)      # step column: host-SIMD-dependent linspace
    sd = _sub(g, "sd.")
    freqs = [float(f) for f in g["freqs"]]
    pk, inf_, target = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"]), torch.as_tensor(g["target"])
    out = tp.render(sd, pk, inf_, None, cobafa_freqs=freqs)
    np.testing.assert_allclose(out.numpy(), g["rendered"], rtol=0, atol=1e-6)
    grads, loss = tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pk, inf_, None, cobafa_freqs=freqs), target))
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-6)
    for k, v in grads.items():
        ref = g["grad." + k]
        np.testing.assert_allclose(v, ref, rtol=1e-4, atol=1e-7 * max(1.0, np.abs(ref).max() / 1e-3), err_msg=k)
    assert len(grads) == sum(1 for k in g if k.startswith("grad."))
