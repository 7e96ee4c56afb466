"""GPU parity of the whole render path (NerfRenderer.forward + backward, reference core.py:209-267)
against G9, which was captured from the reference's own NerfRenderer with the weights kernel
restated by oracle/weights_ref.c."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(DEV)


def build_renderer(g, bg=True):
    from tinynerf_amd import core, models as m
    sd = {k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}
    field = m.KPlanesFeatureField(32)
    field.planes = torch.nn.ModuleList([torch.nn.ModuleList([
        m.KPlanesFeaturePlane(32, tuple(sd[f"feature_module.planes.{s}.0.plane"].shape[2:])) for _ in range(3)]) for s in range(3)])
    r = core.NerfRenderer(field, m.VanillaOpacityDecoder(96), m.VanillaColorDecoder(8, 96, 64, 3), cu(g["bg"]) if bg else None)
    r.load_state_dict(sd)                       # reference checkpoint keys load unchanged
    return r.to(DEV)


@pytest.mark.parametrize("fused", [True, False])
def test_renderer_forward_backward_vs_reference(fused, heads):
    """fused = one autograd node over the C-ABI kernels; not fused = module-by-module like the reference."""
    g = load_golden("G9_renderer_kplanes")
    r = build_renderer(g)
    r.fused = fused
    packed, info = cu(g["packed"]), cu(g["info"], torch.int32)
    out = r(packed, info)
    assert out.shape == (info.shape[0], 3)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["rendered"], rtol=0, atol=TOL)
    loss = torch.nn.functional.mse_loss(out, cu(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    got = {name: p.grad.cpu().numpy() for name, p in r.named_parameters()}
    for name, v in got.items():
        assert v.shape == g["grad." + name].shape, name
    # against the CPU port (pinned to G9 by tests/test_oracle_torch_port.py): 5e-5 of each tensor's largest element (a plane
    # texel sums ~100 atomics in arbitrary order behind a 4-layer chain), up to the state of fp32-tie ReLU units
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    sd = {k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}
    pk, inf_, bgc, tgt = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"]), torch.as_tensor(g["bg"]), torch.as_tensor(g["target"])
    # 83 % of this fixture's samples sit behind a terminated ray: the reference's own sigma-head gradient is determined to 3e-4
    # (weights) ... 9e-4 (first-layer bias) by its fp32 suffix sums (oracle weights_conditioning: exact evaluation + three noise
    # draws); tolerance 4 x that, capped: nothing on this fixture may be looser than 4e-3; with no tie flipped the result is
    # also held against the golden gradients themselves
    assert_grads_match_up_to_relu_ties(got, lambda: tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pk, inf_, bgc), tgt))[0], 5e-5,
                                       weights_conditioning=True, golden={n: g["grad." + n] for n in got}, cond_cap=4e-3)
    # no background colour
    out2 = build_renderer(g, bg=False)(packed, info)
    np.testing.assert_allclose(out2.detach().cpu().numpy(), g["rendered_nobg"], rtol=0, atol=TOL)


def test_renderer_intermediates(heads):
    from tinynerf_amd import core
    g = load_golden("G9_renderer_kplanes")
    r = build_renderer(g)
    packed, info = cu(g["packed"]), cu(g["info"], torch.int32)
    with torch.no_grad():
        sig = r.sigma_decoder(r.feature_module(packed[:, :3])).ravel()
        np.testing.assert_allclose(sig.cpu().numpy(), g["sigma"], rtol=2e-5, atol=TOL)
        w = core.NerfWeights.apply(sig, packed[:, 6].contiguous(), info, 1e-4)
    np.testing.assert_allclose(w.cpu().numpy(), g["weights"], rtol=0, atol=TOL)
    assert int((w == 0).sum()) == pytest.approx(int(g["n_terminated"]), abs=2) and int(g["n_terminated"]) > 0


def test_renderer_empty_iteration(capsys):
    """core.py:235-254: N == 0 (and all-masked) -> background for every ray, backward does not crash."""
    g = load_golden("G9_renderer_kplanes")
    e = load_golden("G9b_renderer_empty")
    r = build_renderer(g)
    R = g["info"].shape[0]
    out = r(torch.zeros((0, 7), device=DEV), torch.zeros((R, 2), dtype=torch.int32, device=DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), e["rendered_empty"], atol=0)
    assert "Empty iteration" in capsys.readouterr().out
    out.sum().backward()
    # all-masked: threshold above 1 terminates every ray before its first sample
    out = r(cu(g["packed"]), cu(g["info"], torch.int32), early_termination_threshold=2.0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), e["rendered_empty"], atol=0)


def test_fused_accumulates_into_existing_grads():
    """harness option: parameter gradients are added in place into param.grad (two steps == twice one step)."""
    g = load_golden("G9_renderer_kplanes")
    packed, info, target = cu(g["packed"]), cu(g["info"], torch.int32), cu(g["target"])
    r1 = build_renderer(g)                                  # one step, gradients returned through autograd
    torch.nn.functional.mse_loss(r1(packed, info), target).backward()
    one = {name: p.grad.clone() for name, p in r1.named_parameters()}
    r = build_renderer(g)
    for p in r.parameters():
        p.grad = torch.zeros_like(p)
    r.accumulate_into_grad = True
    for _ in range(2):
        torch.nn.functional.mse_loss(r(packed, info), target).backward()
    for name, p in r.named_parameters():
        # "twice one step", against the same kernels: equal up to the order of the atomic sums (plane texels, weight tiles)
        own = 2 * one[name].cpu().numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), own, rtol=0, atol=2e-5 * float(np.abs(own).max()), err_msg=name)
        # and against the reference's golden gradients: 1e-4 of the largest element per tensor; the sigma head 1.2e-3 = 4 x the
        # 3e-4 to which the reference's own fp32 weights backward determines it on this fixture (oracle weights_conditioning;
        # the tie-aware comparison of this fixture is test_renderer_forward_backward_vs_reference)
        ref = 2 * g["grad." + name]
        tol = 1.2e-3 if name.startswith("sigma_decoder") else 1e-4
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=0, atol=tol * float(np.abs(ref).max()), err_msg=name)


@pytest.mark.parametrize("n_rays,per_ray", [(37, 29), (300, 113)])
def test_fused_gather_is_bit_identical_to_two_launches(n_rays, per_ray, monkeypatch, heads):
    """tn_kplanes_mlp_fwd_pair / tn_kplanes_mlp_bwd_pair (gather / scatter inside the MLP launches, north star) against
    tn_kplanes_fwd + tn_mlp_fwd_stash_pair and tn_mlp_bwd_pair + tn_kplanes_bwd:
    same arithmetic in the same order -> identical bits for the rendered colours (gradients: the same terms through fp32
    atomics in a different order).  Points partly outside [-1, 1], ragged rays."""
    from tinynerf_amd import core, fused, models as m
    torch.manual_seed(n_rays)
    field = m.KPlanesFeatureField(32, (16, 40, 96))
    r = core.NerfRenderer(field, m.VanillaOpacityDecoder(96), m.VanillaColorDecoder(8, 96, 64, 3), torch.ones(3)).to(DEV)
    cnt = torch.randint(0, per_ray, (n_rays,), dtype=torch.int32)
    info = torch.stack([torch.cumsum(cnt, 0, dtype=torch.int32) - cnt, cnt], -1).to(DEV)
    n = int(cnt.sum())
    packed = torch.rand(n, 7, device=DEV)
    packed[:, :3] = packed[:, :3] * 2.2 - 1.1
    packed[:, 3:6] = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=DEV), dim=-1)[torch.repeat_interleave(
        torch.arange(n_rays, device=DEV), cnt.to(DEV).long())]
    packed[:, 6] = 0.02
    target = torch.rand(n_rays, 3, device=DEV)
    res = {}
    for fuse in (False, True):
        monkeypatch.setattr(fused, "FUSE_GATHER", fuse)
        monkeypatch.setattr(fused, "FUSE_SCATTER", fuse)            # backward twin: tn_kplanes_mlp_bwd_pair
        r.zero_grad(set_to_none=True)
        out = r(packed, info)
        torch.nn.functional.mse_loss(out, target).backward()
        with torch.no_grad():                      # inference form: tn_kplanes_mlp_fwd (gather + sigma head), colour head gated by w
            out_eval = r(packed, info)
        res[fuse] = (out.detach().clone(), {k: p.grad.clone() for k, p in r.named_parameters()}, out_eval.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    np.testing.assert_allclose(res[True][2].cpu().numpy(), res[True][0].cpu().numpy(), rtol=0, atol=1e-6)      # eval == train forward
    for k, g in res[True][1].items():         # every gradient ends in fp32 atomics: same terms, different order
        np.testing.assert_allclose(g.cpu().numpy(), res[False][1][k].cpu().numpy(), rtol=0, atol=2e-6 * float(g.abs().max()), err_msg=k)


@pytest.mark.parametrize("n_rays,per_ray,sigma_bias", [(37, 29, 0.0), (300, 113, 0.0), (300, 113, 6.0), (1, 1, 0.0)])
def test_inference_pair_form_is_bit_identical_to_the_gated_form(n_rays, per_ray, sigma_bias, monkeypatch, heads):
    """Round 4: the inference render has two forms -- gather + sigma head -> weights -> colour head on the live tiles -> composite
    (core.py:239-265 as the reference runs it), and gather + BOTH heads of every sample in one launch (tn_kplanes_mlp_fwd_pair with
    NULL workspaces) -> weights + composite.  A sample with w == 0 contributes exactly 0 in both, every other sample runs the same
    MFMA sequence: the rendered colours must be the same bits, with a thin medium (every sample live) and with a dense one (sigma
    bias + 6: rays terminate, most tiles dead).  The switch follows the live fraction of the previous call without a host sync."""
    from tinynerf_amd import core, fused, models as m
    torch.manual_seed(n_rays + int(sigma_bias))
    field = m.KPlanesFeatureField(32, (16, 40, 96))
    r = core.NerfRenderer(field, m.VanillaOpacityDecoder(96), m.VanillaColorDecoder(8, 96, 64, 3), torch.ones(3)).to(DEV)
    with torch.no_grad():
        r.sigma_decoder.net.net[2].bias += sigma_bias
    cnt = torch.randint(0 if n_rays > 1 else 1, per_ray + 1, (n_rays,), dtype=torch.int32)
    info = torch.stack([torch.cumsum(cnt, 0, dtype=torch.int32) - cnt, cnt], -1).to(DEV)
    n = int(cnt.sum())
    packed = torch.rand(n, 7, device=DEV)
    packed[:, :3] = packed[:, :3] * 2.2 - 1.1
    packed[:, 3:6] = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=DEV), dim=-1)[torch.repeat_interleave(
        torch.arange(n_rays, device=DEV), cnt.to(DEV).long())]
    packed[:, 6] = 0.05
    outs, calls = {}, []
    orig = fused.L.call
    monkeypatch.setattr(fused.L, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    with torch.no_grad():
        for form in (False, True):
            monkeypatch.setattr(fused, "INFER_PAIR", form)
            r.__dict__.pop("_stats", None)                       # a fresh renderer state: no measurement has landed -> the gated form
            del calls[:]
            first = r(packed, info).clone()
            assert "tn_kplanes_mlp_fwd" in calls and "tn_kplanes_mlp_fwd_pair" not in calls, calls
            if form:
                torch.cuda.synchronize()                         # the first call's live fraction has landed
                monkeypatch.setattr(fused, "INFER_PAIR_MIN_LIVE", 0.0)       # (whatever it is: the pair form now)
                del calls[:]
                outs[form] = r(packed, info).clone()
                assert "tn_kplanes_mlp_fwd_pair" in calls and "tn_kplanes_mlp_fwd" not in calls, calls
                monkeypatch.setattr(fused, "INFER_PAIR_MIN_LIVE", 0.6)
                assert torch.equal(first, outs[form])
            else:
                outs[form] = first
        assert torch.equal(outs[True], outs[False])
        # the switch: the next call follows the live fraction of the most recent call whose measurement has landed
        torch.cuda.synchronize()
        fused._infer_prefers_pair(r._stats)
        live = r._stats["infer_live"]["value"]
        with torch.no_grad():
            w_live = float((core.NerfWeights.apply(r.sigma_decoder(r.feature_module(packed[:, :3])).ravel(), packed[:, 6].contiguous(), info, 1e-4) > 0)
                           .float().mean()) if n else 1.0
        assert abs(live - w_live) < 1e-6
        del calls[:]
        out3 = r(packed, info)
        assert ("tn_kplanes_mlp_fwd_pair" in calls) == (live >= fused.INFER_PAIR_MIN_LIVE), (live, calls)
        assert torch.equal(out3, outs[False])
    if sigma_bias > 0:
        assert live < 0.5, live                                  # the dense medium really is mostly dead
