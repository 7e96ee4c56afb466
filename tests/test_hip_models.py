"""GPU parity tests of the field models (tinynerf_amd.models -> fused MFMA MLP / K-Planes kernels)
against golden vectors captured from the reference and against the CPU oracle.  fp32, tolerance
1e-5 absolute (north star) unless a comment says otherwise."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tinynerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(DEV)


def sub(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(v) for k, v in g.items() if k.startswith(prefix)}


def models():
    from tinynerf_amd import models as m
    return m


def test_state_dict_keys_match_reference():
    """key names / shapes of the reference checkpoints."""
    m = models()
    from tinynerf_amd import core
    r = core.NerfRenderer(m.KPlanesFeatureField(32), m.VanillaOpacityDecoder(96), m.VanillaColorDecoder(8, 96, 64, 3))
    sd = r.state_dict()
    assert sd["feature_module.planes.2.1.plane"].shape == (1, 32, 512, 512)
    assert sd["sigma_decoder.net.net.0.weight"].shape == (64, 96) and sd["sigma_decoder.net.net.2.bias"].shape == (1,)
    assert sd["rgb_decoder.pe.freqs"].shape == (8,)
    assert [k for k in sd if k.startswith("rgb_decoder.net.net.") and k.endswith("weight")] == \
        [f"rgb_decoder.net.net.{i}.weight" for i in ("0", "2.0", "3.0", "4.0", "5")]
    v = m.VanillaFeatureMLP(10, 256, 8)
    assert v.state_dict()["encoding.freqs"].shape == (10,) and v.state_dict()["net.net.10.weight"].shape == (256, 256)
    assert sum(p.numel() for p in m.KPlanesFeatureField(32).parameters()) == 33030144
    assert sum(p.numel() for p in v.parameters()) == 607744


def test_posenc():
    m = models()
    g = load_golden("G6_posenc")
    for F in (3, 8, 10):
        pe = m.PositionalEncoding(F).to(DEV)
        assert np.array_equal(pe.freqs.cpu().numpy(), g[f"freqs{F}"])
        np.testing.assert_allclose(pe(cu(g["x"])).cpu().numpy(), g[f"enc{F}"], rtol=0, atol=TOL)
    assert m.PositionalEncoding(4).to(DEV)(torch.zeros(2, 3, 5, 3, device=DEV)).shape == tuple(g["enc4"])


def test_vanilla_heads_forward(heads):
    m = models()
    g = load_golden("G8_vanilla_heads")
    fm = m.VanillaFeatureMLP(6, 64, 3); od = m.VanillaOpacityDecoder(64); cd = m.VanillaColorDecoder(8, 64, 64, 3)
    fm.load_state_dict(sub(g, "fm.")); od.load_state_dict(sub(g, "od.")); cd.load_state_dict(sub(g, "cd."))
    fm.to(DEV); od.to(DEV); cd.to(DEV)
    with torch.no_grad():
        feat = fm(cu(g["x"]))
        np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], rtol=0, atol=TOL)
        np.testing.assert_allclose(od(cu(g["feat"])).cpu().numpy(), g["sigma"], rtol=1e-5, atol=TOL)
        np.testing.assert_allclose(cd(cu(g["feat"]), cu(g["dirs"])).cpu().numpy(), g["rgb"], rtol=0, atol=TOL)


def test_decoders_96_forward(heads):
    m = models()
    g = load_golden("G8b_decoders_96")
    od = m.VanillaOpacityDecoder(96); cd = m.VanillaColorDecoder(8, 96, 64, 3)
    od.load_state_dict(sub(g, "od.")); cd.load_state_dict(sub(g, "cd.")); od.to(DEV); cd.to(DEV)
    with torch.no_grad():
        s = od(cu(g["feat"])); c = cd(cu(g["feat"]), cu(g["dirs"]))
    assert s.shape == (200, 1) and c.shape == (200, 3)
    np.testing.assert_allclose(s.cpu().numpy(), g["sigma"], rtol=1e-5, atol=TOL)
    np.testing.assert_allclose(c.cpu().numpy(), g["rgb"], rtol=0, atol=TOL)


@pytest.mark.parametrize("hidden,layers,n", [(256, 8, 1000), (128, 5, 333), (32, 2, 64), (64, 0, 31)])
def test_mlp_shapes_vs_oracle(hidden, layers, n):
    """reference tests/test_models.py:6-20 shapes + values vs the oracle for every hidden width,
    including the width-256 stack whose weights are streamed from L2 instead of LDS."""
    m = models()
    torch.manual_seed(hidden)
    fm = m.VanillaFeatureMLP(10, hidden, layers)
    sd = {k: v.numpy() for k, v in fm.state_dict().items()}
    x = torch.rand(n, 3) * 2 - 1
    ref = orc.vanilla_features(x.numpy(), orc.mlp_layers(sd, "net.net."), 10)
    fm.to(DEV)
    with torch.no_grad():
        out = fm(x.to(DEV))
    assert out.shape == (n, hidden)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL)


def test_mlp_generic_out_widths():
    """plain MLP with an output width that is neither <= 4 nor a multiple of 32 (KPlanesExplicitColorDecoder: 3*C)."""
    m = models()
    torch.manual_seed(1)
    net = m.MLP(40, 64, 1, 72)
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    x = torch.randn(130, 40)
    ref = orc.mlp_forward(x.numpy(), orc.mlp_layers(sd, "net."))
    net.to(DEV)
    with torch.no_grad():
        out = net(x.to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL)


def _small_field(g):
    m = models()
    field = m.KPlanesFeatureField(32)
    field.planes = torch.nn.ModuleList([torch.nn.ModuleList([
        m.KPlanesFeaturePlane(32, tuple(g[f"plane_{s}_0"].shape[2:])) for _ in range(3)]) for s in range(3)])
    with torch.no_grad():
        for s in range(3):
            for p in range(3):
                field.planes[s][p].plane.copy_(torch.as_tensor(g[f"plane_{s}_{p}"]))
    return field.to(DEV)


def test_kplanes_plane_arange():
    m = models()
    g = load_golden("G7a_plane_arange")
    pl = m.KPlanesFeaturePlane(8, (3, 5)).to(DEV)
    with torch.no_grad():
        pl.plane.copy_(cu(g["plane"]).expand(1, 8, 3, 5))
        out = pl(cu(g["xy"]))
    assert out.shape == (8, 8)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), g["out"][:, 0], rtol=0, atol=1e-6)


def test_kplanes_field_fwd_bwd():
    g = load_golden("G7b_kplanes_field")
    field = _small_field(g)
    assert field.planes[0][0].plane.shape == (1, 32, 8, 8)
    feat = field(cu(g["x"]))
    assert feat.shape == (256, 96)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat"], rtol=0, atol=TOL)
    feat.backward(cu(g["grad_feat"]))
    for s in range(3):
        for p in range(3):
            got = field.planes[s][p].plane.grad
            assert got.shape == g[f"grad_plane_{s}_{p}"].shape
            np.testing.assert_allclose(got.cpu().numpy(), g[f"grad_plane_{s}_{p}"], rtol=1e-5, atol=2e-5)   # sums of ~100 atomics
    np.testing.assert_allclose(field.loss_tv().item(), float(g["loss_tv"]), rtol=1e-5)
    np.testing.assert_allclose(field.loss_l1().item(), float(g["loss_l1"]), rtol=1e-5)


def test_kplanes_full_resolution_vs_oracle():
    """config-3 planes (128/256/512, 126 MiB) on 4096 points, oracle finishes in seconds; also reads
    coordinates straight out of a packed [N,7] tensor (row stride 7)."""
    m = models()
    torch.manual_seed(0)
    field = m.KPlanesFeatureField(32)
    planes = [[field.planes[s][p].plane.detach().numpy() for p in range(3)] for s in range(3)]
    packed = torch.rand(4096, 7) * 2 - 1
    ref = orc.kplanes_features(packed[:, :3].numpy(), planes)
    field.to(DEV)
    with torch.no_grad():
        out = field(packed.to(DEV)[:, :3])
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=TOL)


def test_cobafa_forward(matmul):
    """tn_cobafa_fwd against the reference's own output (G11): HIP gathers -> oracle MLP (the golden's MLP is 16
    wide, below the MFMA kernel's 32) must reproduce the captured features; and against the oracle's gathers."""
    m = models()
    g = load_golden("G11_cobafa")
    sd = {k: np.asarray(v) for k, v in sub(g, "sd.").items()}
    freqs = [float(f) for f in g["freqs"]]
    cf = m.CobafaFeatureField(basis_res=[4, 5, 6], coef_res=4, freqs=freqs, channels=[2, 2, 2], mlp_hidden_dim=32)
    with torch.no_grad():
        for i in range(3):
            cf.basis_grids[i].grid.copy_(torch.from_numpy(sd[f"basis_grids.{i}.grid"]))
        cf.coef_grid.grid.copy_(torch.from_numpy(sd["coef_grid.grid"]))
    cf.to(DEV).eval()
    assert cf.coef_grid.grid.is_contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        gathered = cf.features(cu(g["x"])).cpu().numpy()
    basis = [sd[f"basis_grids.{i}.grid"] for i in range(3)]
    np.testing.assert_allclose(gathered, orc.cobafa_gather(g["x"], basis, sd["coef_grid.grid"], freqs), rtol=0, atol=TOL)
    np.testing.assert_allclose(orc.mlp_forward(gathered, orc.mlp_layers(sd, "mlp.net.")), g["feat"], rtol=0, atol=TOL)
    out = cf(cu(g["x"]))
    assert out.shape == (128, 32)


@pytest.mark.parametrize("res,ch,freqs,coef_res", [
    ([32, 51, 70, 89, 108, 128], [8, 8, 8, 4, 4, 4], [2., 3.2, 4.4, 5.6, 6.8, 8.], 64),      # run.py:176-181
    ([7, 12, 9], [2, 5, 8], [1.5, 2.5, 4.], 6),                                               # run-time level count, odd channel counts
])
def test_cobafa_default_config_against_grid_sample(res, ch, freqs, coef_res, matmul):
    """run.py:176-181's Cobafa configuration (6 levels, 36 features) and a small irregular one: forward and every grid gradient
    against ATen's CPU grid_sampler_3d (what the reference runs), points partly outside [-1,1] (zeros padding)."""
    m = models()
    torch.manual_seed(3)
    cf = m.CobafaFeatureField(basis_res=res, coef_res=coef_res, freqs=freqs, channels=ch, mlp_hidden_dim=128).to(DEV)
    x = (torch.rand(20001, 3, device=DEV) * 2.4 - 1.2)
    feat = cf.features(x)
    gfeat = torch.randn_like(feat)
    (feat * gfeat).sum().backward()
    got = {n: p.grad.clone() for n, p in cf.named_parameters() if p.grad is not None}

    def lookup(grid, pts):
        return torch.nn.functional.grid_sample(grid, pts.view(1, -1, 1, 1, 3), align_corners=True).view(grid.size(1), -1).t()

    leaves = {n: p.detach().cpu().contiguous().clone().requires_grad_(True) for n, p in cf.named_parameters() if "grid" in n}
    xc = x.cpu()
    coefs = lookup(leaves["coef_grid.grid"], xc)
    ref = torch.cat([lookup(leaves[f"basis_grids.{i}.grid"], 2. * ((f * xc) % 1.) - 1.) * coefs[:, [i]] for i, f in enumerate(freqs)], -1)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=TOL)
    (ref * gfeat.cpu()).sum().backward()
    for n, leaf in leaves.items():
        r = leaf.grad.cpu().numpy()
        np.testing.assert_allclose(got[n].cpu().numpy(), r, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(r).max()), err_msg=n)
    # single-grid module
    y = cf.coef_grid(x[:64])
    np.testing.assert_allclose(y.detach().cpu().numpy(), lookup(leaves["coef_grid.grid"], xc[:64]).detach().numpy(), rtol=0, atol=TOL)
    # empty input
    assert cf.features(x[:0]).shape == (0, sum(ch))


def test_cobafa_renderer_trains():
    """method == cobafa (run.py:176-181) through build_renderer: optimisation steps run and the loss moves down."""
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=2, res=48, seed=3, device="cpu")
    cfg = TrainConfig(method="cobafa", batch_size=256, n_samples=64, occupancy_res=32, deterministic=True)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    losses = []
    for _ in range(6):
        tr.step()
        losses.append(tr.loss_value())
    assert all(np.isfinite(l) for l in losses) and min(losses[1:]) < losses[0], losses


def _grads(module):
    return {n: p.grad.detach().cpu().numpy() for n, p in module.named_parameters()}


def _port_leaves(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("freqs") else v) for k, v in sd.items()}


def test_vanilla_heads_backward(heads):
    """gradients of sigma / rgb w.r.t. every parameter of the feature MLP and the decoders (G8; the golden pins the CPU port,
    tests/test_oracle_golden.py).  2e-5 of each tensor's largest element (sums over 256 samples through <= 9 layers in MFMA
    K-order vs ATen's), up to the state of fp32-tie ReLU units (tests/_ties.py)."""
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    m = models()
    g = load_golden("G8_vanilla_heads")
    fm = m.VanillaFeatureMLP(6, 64, 3); od = m.VanillaOpacityDecoder(64); cd = m.VanillaColorDecoder(8, 64, 64, 3)
    fm.load_state_dict(sub(g, "fm.")); od.load_state_dict(sub(g, "od.")); cd.load_state_dict(sub(g, "cd."))
    fm.to(DEV); od.to(DEV); cd.to(DEV)
    x, dirs = cu(g["x"]), cu(g["dirs"])
    sd = {**{"fm." + k: v for k, v in sub(g, "fm.").items()}, **{"od." + k: v for k, v in sub(g, "od.").items()},
          **{"cd." + k: v for k, v in sub(g, "cd.").items()}}
    xc, dc = torch.as_tensor(g["x"]), torch.as_tensor(g["dirs"])

    def ref(head):
        def run():
            lv = _port_leaves(sd)
            feat = tp.mlp(lv, "fm.net.net.", tp.posenc(xc, lv["fm.encoding.freqs"]))
            if head == "sigma":
                y = tp._TruncExp.apply(tp.mlp(lv, "od.net.net.", feat) - 1.)
                (y * torch.as_tensor(g["grad_sigma"])).sum().backward()
            else:
                y = torch.sigmoid(tp.mlp(lv, "cd.net.net.", torch.cat([tp.posenc(dc, lv["cd.pe.freqs"]), dc, feat], -1)))
                (y * torch.as_tensor(g["grad_rgb"])).sum().backward()
            return {k: v.grad.numpy() for k, v in lv.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
        return run
    sig = od(fm(x))
    (sig * cu(g["grad_sigma"])).sum().backward()
    got = {**{"fm." + k: v for k, v in _grads(fm).items()}, **{"od." + k: v for k, v in _grads(od).items()}}
    for k, v in got.items():
        assert v.shape == g["gsig." + k].shape
    assert_grads_match_up_to_relu_ties(got, ref("sigma"), 2e-5)
    fm.zero_grad(); od.zero_grad()
    rgb = cd(fm(x), dirs)
    (rgb * cu(g["grad_rgb"])).sum().backward()
    got = {**{"fm." + k: v for k, v in _grads(fm).items()}, **{"cd." + k: v for k, v in _grads(cd).items()}}
    assert_grads_match_up_to_relu_ties(got, ref("rgb"), 2e-5)


def test_decoders_96_backward(heads):
    """K-Planes-shaped heads: grads w.r.t. parameters AND the incoming features (G8b), 2e-5 of each tensor's largest element."""
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    m = models()
    g = load_golden("G8b_decoders_96")
    od = m.VanillaOpacityDecoder(96); cd = m.VanillaColorDecoder(8, 96, 64, 3)
    od.load_state_dict(sub(g, "od.")); cd.load_state_dict(sub(g, "cd.")); od.to(DEV); cd.to(DEV)
    feat = cu(g["feat"]).requires_grad_(True)
    s = od(feat); c = cd(feat, cu(g["dirs"]))
    ((s * cu(g["grad_sigma"])).sum() + (c * cu(g["grad_rgb"])).sum()).backward()
    got = {"feat": feat.grad.cpu().numpy(), **{"od." + k: v for k, v in _grads(od).items()}, **{"cd." + k: v for k, v in _grads(cd).items()}}
    assert got["feat"].shape == g["grad_feat"].shape
    sd = {**{"od." + k: v for k, v in sub(g, "od.").items()}, **{"cd." + k: v for k, v in sub(g, "cd.").items()}}
    dc = torch.as_tensor(g["dirs"])

    def ref():
        lv = _port_leaves(sd)
        f = torch.as_tensor(g["feat"]).requires_grad_(True)
        so = tp._TruncExp.apply(tp.mlp(lv, "od.net.net.", f) - 1.)
        co = torch.sigmoid(tp.mlp(lv, "cd.net.net.", torch.cat([tp.posenc(dc, lv["cd.pe.freqs"]), dc, f], -1)))
        ((so * torch.as_tensor(g["grad_sigma"])).sum() + (co * torch.as_tensor(g["grad_rgb"])).sum()).backward()
        return {"feat": f.grad.numpy(), **{k: v.grad.numpy() for k, v in lv.items() if isinstance(v, torch.Tensor) and v.requires_grad}}
    assert_grads_match_up_to_relu_ties(got, ref, 2e-5)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_mlp_backward_ragged_sizes_vs_torch(seed):
    """n not a multiple of 32, several tiles per wave: the fused backward against torch autograd of the same fp32 network
    evaluated with torch ops on the device (rocBLAS), 2e-5 of each tensor's largest element, any seed (tests/_ties.py)."""
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    m = models()
    torch.manual_seed(seed)
    net = m.MLP(40, 64, 2, 3).to(DEV)
    x = torch.randn(1000 + 17, 40, device=DEV, requires_grad=True)
    gy = torch.randn(1017, 3, device=DEV)
    y = net(x)
    y.backward(gy)
    got = {"x": x.grad.cpu().numpy(), **{n: p.grad.cpu().numpy() for n, p in net.named_parameters()}}
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    y_ref = []

    def ref():
        lv = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xl = x.detach().clone().requires_grad_(True)
        y2 = tp.mlp(lv, "net.", xl)
        y_ref.append(y2.detach())
        y2.backward(gy)
        return {"x": xl.grad.cpu().numpy(), **{k: v.grad.cpu().numpy() for k, v in lv.items()}}
    assert_grads_match_up_to_relu_ties(got, ref, 2e-5)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref[0].cpu().numpy(), atol=TOL)


def test_plane_regularisers_fwd_bwd_vs_torch():
    """a19: fused TV / L1 passes against torch autograd of the reference formulas (models.py:115-121)."""
    m = models()
    torch.manual_seed(5)
    field = m.KPlanesFeatureField(32)
    field.planes = torch.nn.ModuleList([torch.nn.ModuleList([m.KPlanesFeaturePlane(32, r) for _ in range(3)])
                                        for r in ((8, 8), (12, 10), (33, 17))]).to(DEV)
    field.to(DEV)
    loss = field.loss_tv() * 0.7 + field.loss_l1() * 0.3
    loss.backward()
    got = [p.plane.grad.clone() for s in field.planes for p in s]
    ref_params = [p.plane.detach().clone().requires_grad_(True) for s in field.planes for p in s]
    tv = sum(torch.nn.functional.mse_loss(p[:, :, 1:, :], p[:, :, :-1, :]) + torch.nn.functional.mse_loss(p[:, :, :, 1:], p[:, :, :, :-1]) for p in ref_params) / 9
    l1 = sum(p.abs().mean() for p in ref_params) / 9
    ref = tv * 0.7 + l1 * 0.3
    ref.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-6)
    for g, p in zip(got, ref_params):
        np.testing.assert_allclose(g.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(field.regulariser(1e-4, 0.0).item(), 1e-4 * tv.item(), rtol=1e-6)
    # harness form (tn_plane_reg_multi): value + upstream-scaled gradient added to the existing .grad, one launch
    before = [p.plane.grad.clone() for s in field.planes for p in s]
    val = field.regulariser_step(0.7, 0.3, upstream=1024.0)
    np.testing.assert_allclose(val.item(), ref.item(), rtol=1e-6)
    for b, pl, p in zip(before, [p.plane for s in field.planes for p in s], ref_params):
        want = b.cpu().numpy() + 1024.0 * p.grad.cpu().numpy()
        np.testing.assert_allclose(pl.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", [
    dict(kind="vanilla", F=10, H=256, layers=8, n=700),        # reference config (run.py:131): 60->256 x9 ->256
    dict(kind="vanilla", F=6, H=128, layers=5, n=333),
    dict(kind="mlp", K=36, H=128, layers=5, out=128, n=500),   # Cobafa MLP (run.py:141-147)
    dict(kind="mlp", K=40, H=64, layers=6, out=3, n=257),      # deeper than the register-resident forms cover
    dict(kind="color", F=4, dim=256, H=128, layers=1, n=300),  # reference tests/test_core.py:58 decoder shape
    dict(kind="vanilla", F=10, H=256, layers=2, n=40037),      # more 32-sample tiles than workgroups / waves: the persistent
    dict(kind="mlp", K=36, H=128, layers=2, out=128, n=40037), # loops and their next-tile prefetches run several rounds
    dict(kind="mlp", K=36, H=128, layers=2, out=40, n=1000),   # output width below H and not a multiple of 32
    dict(kind="mlp", K=24, H=256, layers=3, out=200, n=999),
    dict(kind="mlp", K=147, H=128, layers=3, out=288, n=500),  # output wider than H: KPlanesExplicitColorDecoder(96, 8, 128)
    dict(kind="mlp", K=40, H=64, layers=4, out=100, n=300),
    # edge sizes of the persistent layer kernels (weights in registers, tiles through LDS): one sample, one tile + one sample,
    # fewer tiles than tile streams per workgroup (H = 128 runs two streams), an odd number of tiles
    dict(kind="mlp", K=36, H=128, layers=5, out=128, n=5000),   # Cobafa's stack (models.py:239-247): plain inputs staged as rows (round 4)
    dict(kind="mlp", K=36, H=128, layers=3, out=128, n=1),
    dict(kind="mlp", K=36, H=128, layers=3, out=128, n=33),
    dict(kind="mlp", K=36, H=128, layers=3, out=96, n=65),
    dict(kind="vanilla", F=10, H=256, layers=2, n=1),
    dict(kind="vanilla", F=10, H=256, layers=3, n=33),
])
@pytest.mark.parametrize("stash,seed", [(True, 11), (False, 11), (True, 12), (True, 13)])
def test_wide_deep_mlp_backward_vs_torch(cfg, stash, seed, monkeypatch, matmul):
    """layer-by-layer backward (mlp_bwd_layers.hip) against torch autograd of the same fp32 network on the device, with the
    activations written by the training forward (tn_mlp_fwd_stash) and recomputed by the backward.  ANY seed: hidden units
    whose pre-activation is an fp32 tie may take either ReLU state (tests/_ties.py); everything else must agree to 3e-5 of each
    gradient tensor's largest element (a weight gradient is a sum over n samples of products of ~10-layer-deep quantities,
    accumulated in a different order by the MFMA tiles / atomics than by rocBLAS)."""
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    m = models()
    monkeypatch.setattr(m._FusedMLP, "stash_forward", stash)
    torch.manual_seed(seed)
    n = cfg["n"]
    d = None
    if cfg["kind"] == "vanilla":
        net = m.VanillaFeatureMLP(cfg["F"], cfg["H"], cfg["layers"]).to(DEV)
        x = torch.rand(n, 3, device=DEV) * 2 - 1
        y = net(x)
        prefix = "net.net."
    elif cfg["kind"] == "mlp":
        net = m.MLP(cfg["K"], cfg["H"], cfg["layers"], cfg["out"]).to(DEV)
        x = torch.randn(n, cfg["K"], device=DEV, requires_grad=True)
        y = net(x)
        prefix = "net."
    else:
        net = m.VanillaColorDecoder(cfg["F"], cfg["dim"], cfg["H"], cfg["layers"]).to(DEV)
        x = torch.rand(n, cfg["dim"], device=DEV, requires_grad=True)
        d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV), dim=-1)
        y = net(x, d)
        prefix = "net.net."
    gy = torch.randn_like(y)
    y.backward(gy)
    got = {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}
    if x.requires_grad:
        got["x"] = x.grad.cpu().numpy()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    y_ref = []

    def ref():                       # the same network through torch ops on the device (rocBLAS fp32), ties recorded / forced
        leaves = {k: (v.clone().requires_grad_(True) if not k.endswith("freqs") else v) for k, v in sd.items()}
        xl = x.detach().clone().requires_grad_(x.requires_grad)
        if cfg["kind"] == "vanilla":
            inp = tp.posenc(xl, leaves["encoding.freqs"])
        elif cfg["kind"] == "mlp":
            inp = xl
        else:
            inp = torch.cat([tp.posenc(d, leaves["pe.freqs"]), d, xl], -1)
        y2 = tp.mlp(leaves, prefix, inp)
        if cfg["kind"] == "color":
            y2 = torch.sigmoid(y2)
        y_ref.append(y2.detach())
        y2.backward(gy)
        out = {k: v.grad.cpu().numpy() for k, v in leaves.items() if v.requires_grad}
        if xl.requires_grad:
            out["x"] = xl.grad.cpu().numpy()
        return out
    assert_grads_match_up_to_relu_ties(got, ref, 3e-5)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref[0].cpu().numpy(), atol=2e-5)


def test_fused_adam_matches_torch_adam():
    """tn_adam_multi (FusedAdam) and tn_adam_step against torch.optim.Adam (reference run.py:186 settings) over several
    steps, including a channels_last parameter and the LR scheduler."""
    from tinynerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(1, 32, 16, 16), (64, 147), (64,), (3, 64), (1,)]
    p_ref = [torch.rand(s, device=DEV) for s in shapes]
    p_ref[0] = p_ref[0].contiguous(memory_format=torch.channels_last)
    p_new = [torch.nn.Parameter(p.clone(memory_format=torch.preserve_format)) for p in p_ref]
    p_ref = [torch.nn.Parameter(p) for p in p_ref]
    o_ref = torch.optim.Adam(p_ref, lr=1e-2, eps=1e-15, weight_decay=1e-5)
    o_new = FusedAdam(p_new, lr=1e-2, eps=1e-15, weight_decay=1e-5, zero_grad_in_step=True)
    s_ref = torch.optim.lr_scheduler.MultiStepLR(o_ref, milestones=[3, 5], gamma=0.33)
    s_new = torch.optim.lr_scheduler.MultiStepLR(o_new, milestones=[3, 5], gamma=0.33)
    for it in range(8):
        for a, b in zip(p_ref, p_new):
            g = torch.randn_like(a) * 1024
            a.grad = g.clone(memory_format=torch.preserve_format)
            b.grad = g.clone(memory_format=torch.preserve_format)
        o_ref.step(); o_new.step(); s_ref.step(); s_new.step()
        for a, b in zip(p_ref, p_new):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
            assert float(b.grad.abs().max()) == 0.0          # zeroed in the same pass
    # single-tensor entry point: one more step on a fresh pair
    import ctypes as C
    from tinynerf_amd import _lib as L
    a = torch.nn.Parameter(torch.rand(1000, device=DEV)); b = a.detach().clone()
    g = torch.randn(1000, device=DEV); a.grad = g.clone()
    torch.optim.Adam([a], lr=1e-2, eps=1e-15, weight_decay=1e-5).step()
    mm, vv = torch.zeros_like(b), torch.zeros_like(b)
    L.call("tn_adam_step", b.device, L.ptr(b), L.ptr(g), L.ptr(mm), L.ptr(vv), C.c_int64(1000), C.c_float(1e-2), C.c_float(0.9),
           C.c_float(0.999), C.c_float(1e-15), C.c_float(1e-5), C.c_int32(1), C.c_int32(0))
    np.testing.assert_allclose(b.cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("head", ["sigma", "rgb"])
def test_mlp_stashed_forward_and_ray_table_match_recompute_path(head, heads):
    """Training forward with activation stash (tn_mlp_fwd_stash + TN_MLP_STASHED) and the per-ray aux table
    (TN_ENC_AUX_CAT + tn_dir_encode) against the recompute path with per-sample directions (TN_ENC_DIR_CAT):
    same outputs, same parameter gradients, same grad_x (incl. TN_MLP_ACCUM_GRAD_X)."""
    import ctypes as C
    from tinynerf_amd import _lib as L
    from tinynerf_amd.models import _mlp_desc
    m = models()
    torch.manual_seed(11)
    n, R, F = 4133, 97, 96                                   # ragged: not a multiple of 32
    dev = torch.device(DEV)
    if head == "sigma":
        net = m.VanillaOpacityDecoder(F).to(dev).net
        params = [p.detach().contiguous() for p in net.params()]
        enc_a, enc_b, nf, act, freqs, out = L.ENC_NONE, L.ENC_NONE, 0, L.ACT_EXP_M1, None, 1
    else:
        cd = m.VanillaColorDecoder(8, F, 64, 3).to(dev)
        params = [p.detach().contiguous() for p in cd.net.params()]
        enc_a, enc_b, nf, act, freqs, out = L.ENC_DIR_CAT, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs, 3
    x = torch.rand(n, F, device=dev)
    counts = torch.randint(0, 90, (R,), device=dev)
    counts[-1] += n - counts.sum() if counts.sum() < n else 0
    while int(counts.sum()) > n:
        counts[int(torch.argmax(counts))] -= min(int(counts.sum()) - n, int(counts.max()))
    ray_ids = torch.repeat_interleave(torch.arange(R, dtype=torch.int32, device=dev), counts, output_size=n)
    dirs_ray = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    dirs = dirs_ray[ray_ids.long()].contiguous()
    table = torch.full((R, 56), float("nan"), device=dev)
    if head == "rgb":
        L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(freqs), C.c_int(8), L.ptr(table), C.c_int(56))
        ref_tab = torch.cat([torch.sin(dirs_ray[..., None] * freqs), torch.cos(dirs_ray[..., None] * freqs)], -1).flatten(-2)
        np.testing.assert_allclose(table[:, :48].cpu().numpy(), ref_tab.cpu().numpy(), rtol=0, atol=2e-6)
        assert torch.equal(table[:, 48:51], dirs_ray) and float(table[:, 51:].abs().max()) == 0.0
    gy = torch.randn(n, out, device=dev)

    def run(enc, aux, idx, stash, accum):
        flags = (L.MLP_ACCUM_GRAD_X if accum else 0)
        d = _mlp_desc(params, F, enc, nf, act, freqs, flags, idx, 56 if enc == L.ENC_AUX_CAT else 0)
        fn = L.lib().tn_mlp_bwd_workspace_bytes
        fn.restype = C.c_int64
        nbytes = int(fn(C.byref(d), C.c_int64(n)))
        assert nbytes > 0
        ws = torch.empty(nbytes // 4, device=dev)
        y = torch.empty(n, out, device=dev)
        if stash:
            L.call("tn_mlp_fwd_stash", dev, C.byref(d), L.ptr(x), L.ptr(aux), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(nbytes))
            d.flags = flags | L.MLP_STASHED
        else:
            L.call("tn_mlp_fwd", dev, C.byref(d), L.ptr(x), L.ptr(aux), C.c_int64(n), L.ptr(y), C.c_void_p(None))
        gs = [torch.zeros_like(p) for p in params]
        nl = len(params) // 2
        gw = (C.c_void_p * nl)(*[g.data_ptr() for g in gs[0::2]])
        gb = (C.c_void_p * nl)(*[g.data_ptr() for g in gs[1::2]])
        gx = torch.ones(n, F, device=dev) if accum else torch.empty(n, F, device=dev)
        L.call("tn_mlp_bwd", dev, C.byref(d), L.ptr(x), L.ptr(aux), L.ptr(gy), C.c_int64(n), gw, gb, L.ptr(gx), L.ptr(ws), C.c_int64(nbytes))
        return y, gs, gx

    aux_a = dirs if head == "rgb" else None
    aux_b = table if head == "rgb" else None
    y0, g0, gx0 = run(enc_a, aux_a, None, False, False)
    for stash, accum in ((True, False), (False, True), (True, True)):
        y1, g1, gx1 = run(enc_b, aux_b, ray_ids if head == "rgb" else None, stash, accum)
        np.testing.assert_allclose(y1.cpu().numpy(), y0.cpu().numpy(), rtol=0, atol=2e-6)
        for a_, b_ in zip(g1, g0):
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-5 * max(1.0, float(b_.abs().max())))
        np.testing.assert_allclose((gx1 - (1.0 if accum else 0.0)).cpu().numpy(), gx0.cpu().numpy(), rtol=1e-4, atol=2e-6)


def test_mlp_bwd_pair_matches_two_single_head_backwards(heads):
    """tn_mlp_bwd_pair (colour head + 2-layer sigma head sharing x, one data-gradient pass, grad_x written once)
    against tn_mlp_bwd of the colour head followed by tn_mlp_bwd of the sigma head with TN_MLP_ACCUM_GRAD_X."""
    import ctypes as C
    from tinynerf_amd import _lib as L
    from tinynerf_amd.models import _mlp_desc
    m = models()
    torch.manual_seed(21)
    n, R, F = 5003, 61, 96
    dev = torch.device(DEV)
    sp = [p.detach().contiguous() for p in m.VanillaOpacityDecoder(F).to(dev).net.params()]
    cd = m.VanillaColorDecoder(8, F, 64, 3).to(dev)
    rp = [p.detach().contiguous() for p in cd.net.params()]
    x = torch.rand(n, F, device=dev)
    ray_ids = torch.sort(torch.randint(0, R, (n,), device=dev, dtype=torch.int32)).values.contiguous()
    dirs_ray = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    table = torch.empty(R, 56, device=dev)
    L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(cd.pe.freqs), C.c_int(8), L.ptr(table), C.c_int(56))
    g_rgb, g_sig = torch.randn(n, 3, device=dev), torch.randn(n, 1, device=dev)
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64

    def fwd(params, enc, nf, act, freqs, aux, idx, out):
        d = _mlp_desc(params, F, enc, nf, act, freqs, 0, idx, 56 if enc == L.ENC_AUX_CAT else 0)
        nb = int(fn(C.byref(d), C.c_int64(n)))
        ws = torch.empty(nb // 4, device=dev)
        y = torch.empty(n, out, device=dev)
        L.call("tn_mlp_fwd_stash", dev, C.byref(d), L.ptr(x), L.ptr(aux), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(nb))
        d.flags = L.MLP_STASHED
        return d, ws, nb

    def grads(params):
        gs = [torch.zeros_like(p) for p in params]
        k = len(params) // 2
        return gs, (C.c_void_p * k)(*[g.data_ptr() for g in gs[0::2]]), (C.c_void_p * k)(*[g.data_ptr() for g in gs[1::2]])

    rd, wr, nbr = fwd(rp, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs, table, ray_ids, 3)
    sd, wsg, nbs = fwd(sp, L.ENC_NONE, 0, L.ACT_EXP_M1, None, None, None, 1)
    # reference: two launches
    gr0, gwr, gbr = grads(rp); gs0, gws, gbs = grads(sp)
    gx0 = torch.empty(n, F, device=dev)
    L.call("tn_mlp_bwd", dev, C.byref(rd), L.ptr(x), L.ptr(table), L.ptr(g_rgb), C.c_int64(n), gwr, gbr, L.ptr(gx0), L.ptr(wr.clone()), C.c_int64(nbr))
    sd.flags = L.MLP_STASHED | L.MLP_ACCUM_GRAD_X
    L.call("tn_mlp_bwd", dev, C.byref(sd), L.ptr(x), C.c_void_p(None), L.ptr(g_sig), C.c_int64(n), gws, gbs, L.ptr(gx0), L.ptr(wsg.clone()), C.c_int64(nbs))
    # pair
    sd.flags = L.MLP_STASHED
    gr1, gwr, gbr = grads(rp); gs1, gws, gbs = grads(sp)
    gx1 = torch.full((n, F), float("nan"), device=dev)
    L.call("tn_mlp_bwd_pair", dev, C.byref(rd), C.byref(sd), L.ptr(x), L.ptr(table), L.ptr(g_rgb), L.ptr(g_sig), C.c_int64(n), gwr, gbr, gws, gbs,
           L.ptr(gx1), L.ptr(wr), C.c_int64(nbr), L.ptr(wsg), C.c_int64(nbs))
    np.testing.assert_allclose(gx1.cpu().numpy(), gx0.cpu().numpy(), rtol=1e-5, atol=2e-6)
    for a_, b_ in zip(gr1 + gs1, gr0 + gs0):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-5 * max(1.0, float(b_.abs().max())))


def test_mlp_pair_full_size_properties():
    """BASELINE config 3's size (2^20 + 13 packed samples, K-Planes heads): (i) rows of the two-head training forward equal a
    plain torch fp32 evaluation of the same modules on a random subset; (ii) parameter gradients are additive over a split of
    the batch -- bwd(all) == bwd(first part) + bwd(rest) -- which no per-tile or per-workgroup bookkeeping error survives."""
    import ctypes as C
    from tinynerf_amd import _lib as L
    from tinynerf_amd.models import _mlp_desc
    m = models()
    torch.manual_seed(5)
    n, R, F = (1 << 20) + 13, 22579, 96
    dev = torch.device(DEV)
    sig = m.VanillaOpacityDecoder(F).to(dev)
    cd = m.VanillaColorDecoder(8, F, 64, 3).to(dev)
    sp = [p.detach().contiguous() for p in sig.net.params()]
    rp = [p.detach().contiguous() for p in cd.net.params()]
    x = torch.rand(n, F, device=dev)
    ray_ids = torch.sort(torch.randint(0, R, (n,), device=dev, dtype=torch.int32)).values.contiguous()
    dirs_ray = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    table = torch.empty(R, 56, device=dev)
    L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(cd.pe.freqs), C.c_int(8), L.ptr(table), C.c_int(56))
    g_rgb, g_sig = torch.randn(n, 3, device=dev), torch.randn(n, 1, device=dev)
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64

    def run(lo, hi):
        k = hi - lo
        xs, ids, gr, gs_ = x[lo:hi], ray_ids[lo:hi].contiguous(), g_rgb[lo:hi], g_sig[lo:hi]
        rd = _mlp_desc(rp, F, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs, 0, ids, 56)
        sd = _mlp_desc(sp, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, 0, None, 0)
        nbr, nbs = int(fn(C.byref(rd), C.c_int64(k))), int(fn(C.byref(sd), C.c_int64(k)))
        wr, wsg = torch.empty(nbr // 4, device=dev), torch.empty(nbs // 4, device=dev)
        rgb, sigma = torch.empty(k, 3, device=dev), torch.empty(k, 1, device=dev)
        L.call("tn_mlp_fwd_stash_pair", dev, C.byref(rd), C.byref(sd), L.ptr(xs), L.ptr(table), C.c_int64(k), L.ptr(rgb), L.ptr(sigma),
               L.ptr(wr), C.c_int64(nbr), L.ptr(wsg), C.c_int64(nbs))
        rd.flags = sd.flags = L.MLP_STASHED
        grs, gss = [torch.zeros_like(p) for p in rp], [torch.zeros_like(p) for p in sp]
        arr = lambda gs, o: (C.c_void_p * (len(gs) // 2))(*[g.data_ptr() for g in gs[o::2]])
        gx = torch.empty(k, F, device=dev)
        L.call("tn_mlp_bwd_pair", dev, C.byref(rd), C.byref(sd), L.ptr(xs), L.ptr(table), L.ptr(gr), L.ptr(gs_), C.c_int64(k),
               arr(grs, 0), arr(grs, 1), arr(gss, 0), arr(gss, 1), L.ptr(gx), L.ptr(wr), C.c_int64(nbr), L.ptr(wsg), C.c_int64(nbs))
        return rgb, sigma, grs + gss, gx

    rgb, sigma, g_all, gx = run(0, n)
    pick = torch.randint(0, n, (8192,), device=dev)
    pick[-1] = n - 1
    with torch.no_grad():
        d = dirs_ray[ray_ids[pick].long()]
        pe = torch.cat([torch.sin(d[..., None] * cd.pe.freqs), torch.cos(d[..., None] * cd.pe.freqs)], -1).flatten(-2)
        ref_rgb = torch.sigmoid(cd.net.net(torch.cat([pe, d, x[pick]], -1)))
        ref_sig = torch.exp(sig.net.net(x[pick]) - 1.0)
    np.testing.assert_allclose(rgb[pick].cpu().numpy(), ref_rgb.cpu().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(sigma[pick].cpu().numpy(), ref_sig.cpu().numpy(), rtol=1e-5, atol=TOL)
    cut = 400_007                                           # not a multiple of the 32-sample tile
    _, _, g_a, gx_a = run(0, cut)
    _, _, g_b, gx_b = run(cut, n)
    # Tolerance by construction.  Each gradient entry is a sum of ~10^6 products; its fp32 evaluation error scales with the sum of the
    # ABSOLUTE products, not with the (possibly cancelling) result.  The yardstick is measured, not typed in: the same modules under torch
    # autograd on the device once in fp64 (exact for this purpose) and once in fp32; the HIP launches -- whole batch, and the sum of the
    # two parts -- may be no further from the fp64 gradients than 4 x torch's own fp32 evaluation is, per tensor.
    def torch_grads(dtype):
        sig_t, cd_t = m.VanillaOpacityDecoder(F).to(dev).to(dtype), m.VanillaColorDecoder(8, F, 64, 3).to(dev).to(dtype)
        sig_t.load_state_dict({k: v.to(dtype) for k, v in sig.state_dict().items()})
        cd_t.load_state_dict({k: v.to(dtype) for k, v in cd.state_dict().items()})
        ps = list(cd_t.net.net.parameters()) + list(sig_t.net.net.parameters())
        for lo in range(0, n, 1 << 18):                   # (chunks: the fp64 activations of 10^6 samples would not be small)
            hi = min(n, lo + (1 << 18))
            dd = dirs_ray[ray_ids[lo:hi].long()].to(dtype)
            fr = cd.pe.freqs.to(dtype)
            pe = torch.cat([torch.sin(dd[..., None] * fr), torch.cos(dd[..., None] * fr)], -1).flatten(-2)
            xx = x[lo:hi].to(dtype)
            o_rgb = torch.sigmoid(cd_t.net.net(torch.cat([pe, dd, xx], -1)))
            o_sig = torch.exp(sig_t.net.net(xx) - 1.0)
            ((o_rgb * g_rgb[lo:hi].to(dtype)).sum() + (o_sig * g_sig[lo:hi].to(dtype)).sum()).backward()
        return [p.grad.double() for p in ps]
    ref64, ref32 = torch_grads(torch.float64), torch_grads(torch.float32)
    for t, a_, b_, r64, r32 in zip(g_all, g_a, g_b, ref64, ref32):
        scale = float(r64.abs().max())
        yard = max(1e-6 * scale, 4.0 * float((r32 - r64).abs().max()))
        assert float((t.double() - r64).abs().max()) <= yard, (tuple(t.shape), float((t.double() - r64).abs().max()), yard)
        assert float(((a_ + b_).double() - r64).abs().max()) <= yard, (tuple(t.shape), float(((a_ + b_).double() - r64).abs().max()), yard)
    assert torch.equal(gx[:cut][:cut - cut % 32], gx_a[:cut - cut % 32])          # whole tiles see the same arithmetic
    np.testing.assert_allclose(gx[cut:].cpu().numpy(), gx_b.cpu().numpy(), rtol=1e-5, atol=2e-6)


def test_adam_with_folded_regulariser_matches_two_passes():
    """FusedAdam.step(plane_reg=...) (tn_adam_reg_multi: TV / L1 gradient built from the current planes inside the update,
    new values written to a second buffer and swapped in) against regulariser_step() followed by a plain FusedAdam.step()."""
    from tinynerf_amd.optim import FusedAdam
    m = models()
    torch.manual_seed(9)

    def make():
        torch.manual_seed(9)
        f = m.KPlanesFeatureField(32)
        f.planes = torch.nn.ModuleList([torch.nn.ModuleList([m.KPlanesFeaturePlane(32, r) for _ in range(3)]) for r in ((8, 8), (12, 10), (33, 17))])
        f.to(DEV)
        lin = torch.nn.Linear(7, 5).to(DEV)
        params = list(f.parameters()) + list(lin.parameters())
        for p in params:
            p.grad = torch.zeros_like(p)
        return f, params, FusedAdam(params, lr=1e-2, eps=1e-15, weight_decay=1e-5, zero_grad_in_step=True)

    fa, pa, oa = make()
    fb, pb, ob = make()
    for it in range(4):
        torch.manual_seed(100 + it)
        for x, y in zip(pa, pb):
            g = torch.randn_like(x) * 1024
            x.grad.copy_(g); y.grad.copy_(g)
        sums_a = torch.zeros(9 * 3, dtype=torch.float64, device=DEV)
        coef = fa.regulariser_step(0.7, 0.3, upstream=1024.0, sums=sums_a)
        oa.step()
        spec, coef_b = fb.regulariser_spec(0.7, 0.3)
        sums_b = torch.zeros(9 * 3, dtype=torch.float64, device=DEV)
        ob.step(plane_reg={"spec": spec, "upstream": 1024.0, "sums": sums_b})
        np.testing.assert_allclose(sums_b.cpu().numpy(), sums_a.cpu().numpy(), rtol=1e-6)
        for x, y in zip(pa, pb):
            np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
            assert float(y.grad.abs().max()) == 0.0 and y.is_contiguous(memory_format=torch.channels_last) == x.is_contiguous(memory_format=torch.channels_last)
        for x, y in zip(pa, pb):
            np.testing.assert_allclose(ob.state[y]["exp_avg_sq"].cpu().numpy(), oa.state[x]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)


def test_mlp_forward_row_gate_skips_dead_tiles_only():
    """tn_mlp_desc.row_gate (inference colour head, core.py:246-251): 32-row tiles whose gates are all 0 yield 0 without
    being evaluated; every row with a non-zero gate equals the ungated result bit for bit."""
    import ctypes as C
    from tinynerf_amd import _lib as L
    from tinynerf_amd.models import _mlp_desc
    m = models()
    torch.manual_seed(4)
    n, F = 1000, 96
    dev = torch.device(DEV)
    cd = m.VanillaColorDecoder(8, F, 64, 3).to(dev)
    params = [p.detach().contiguous() for p in cd.net.params()]
    x = torch.rand(n, F, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    gate = torch.rand(n, device=dev)
    gate[64:192] = 0.0            # four dead tiles
    gate[200:210] = 0.0           # dead rows inside a live tile
    gate[992:] = 0.0              # ragged last tile, dead
    d = _mlp_desc(params, F, L.ENC_DIR_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs)
    y0 = torch.empty(n, 3, device=dev)
    L.call("tn_mlp_fwd", dev, C.byref(d), L.ptr(x), L.ptr(dirs), C.c_int64(n), L.ptr(y0), C.c_void_p(None))
    d.row_gate = gate.data_ptr()
    y1 = torch.full((n, 3), float("nan"), device=dev)
    L.call("tn_mlp_fwd", dev, C.byref(d), L.ptr(x), L.ptr(dirs), C.c_int64(n), L.ptr(y1), C.c_void_p(None))
    live = gate != 0
    assert torch.equal(y1[live], y0[live])
    assert float(y1[64:192].abs().max()) == 0.0 and float(y1[992:].abs().max()) == 0.0
    assert torch.equal(y1[200:210], y0[200:210])          # dead rows of a live tile are still evaluated (harmless)


def test_fused_mlp_takes_the_stash_forward_only_while_recording(monkeypatch):
    """training: the forward writes the backward's activation workspace (tn_mlp_fwd_stash: nothing is recomputed); inside
    torch.no_grad() (infer(), the occupancy refresh) the inference forward runs (wide stacks: tn_mlp_fwd_ws, layer kernels with
    two ping-pong row buffers; everything else: tn_mlp_fwd) and no training workspace is allocated.  (Inside
    Function.forward grad mode is always off and needs_input_grad always reports the parameters: the call site decides.)"""
    m = models()
    from tinynerf_amd import _lib as L
    names = []
    orig = L.call
    monkeypatch.setattr(L, "call", lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1])
    net = m.VanillaFeatureMLP(10, 256, 3).to(DEV)
    x = torch.rand(500, 3, device=DEV) * 2 - 1
    y = net(x)
    assert names == ["tn_mlp_fwd_stash"]
    y.sum().backward()
    assert names == ["tn_mlp_fwd_stash", "tn_mlp_bwd"]
    del names[:]
    with torch.no_grad():
        y2 = net(x)
    assert names == ["tn_mlp_fwd_ws"] and torch.equal(y2, y.detach())
    del names[:]
    od = m.VanillaOpacityDecoder(256).to(DEV)
    with torch.no_grad():
        od(y2)
    assert names == ["tn_mlp_fwd"]


# ------------------------------------------------------------------ the one plain Linear of the path (models.py:186)
@pytest.mark.parametrize("n,fin,fout,bias", [(1000, 96, 96, True), (33, 96, 96, True), (4097, 32, 64, False), (777, 50, 7, True),
                                              (64, 1, 1, True), (100_000, 96, 96, True), (500, 128, 128, True), (0, 96, 96, True)])
def test_linear_fwd_bwd_vs_torch(n, fin, fout, bias):
    """tn_linear_fwd / tn_linear_bwd against torch.nn.functional.linear in fp64 on the host: ragged n, widths that are not
    multiples of 32 (zero padding inside the kernel), no bias, the 128 x 128 limit, the empty batch."""
    m = models()
    g = torch.Generator().manual_seed(n + fin)
    x = torch.randn(n, fin, generator=g)
    w = torch.randn(fout, fin, generator=g) / fin ** 0.5
    b = torch.randn(fout, generator=g) if bias else None
    go = torch.randn(n, fout, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = None if b is None else b.double().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(go.double())
    xt, wt = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    bt = None if b is None else cu(b).requires_grad_(True)
    y = m.linear(xt, wt, bt)
    assert y.shape == (n, fout)
    y.backward(cu(go))

    def close(got, ref, k):       # fp32 accumulation over k terms
        ref = ref.detach().float().numpy()
        assert got.shape == ref.shape
        if ref.size == 0:
            return
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=0, atol=4e-7 * max(k, 8) ** 0.5 * max(float(np.abs(ref).max()), 1e-30) + 1e-30)
    close(y, yr, fin)
    close(xt.grad, xr.grad, fout)
    close(wt.grad, wr.grad, max(n, 1))
    if b is not None:
        close(bt.grad, br.grad, max(n, 1))


@pytest.mark.parametrize("n,fin,fout", [(1000, 96, 96), (33, 50, 7)])
def test_linear_bwd_bias_gradient_alone(n, fin, fout):
    """tn_linear_bwd with grad_weight == NULL and grad_bias given (the header: both independently optional): the bias sums must
    arrive, accumulated, through the C ABI itself -- no dummy weight-gradient buffer"""
    from tinynerf_amd import _lib as L
    import ctypes as C
    g = torch.Generator().manual_seed(n)
    w, go = cu(torch.randn(fout, fin, generator=g)), cu(torch.randn(n, fout, generator=g))
    gb = torch.ones(fout, device=DEV)
    L.call("tn_linear_bwd", DEV, C.c_void_p(None), L.ptr(w), L.ptr(go), C.c_int64(n), C.c_int32(fin), C.c_int32(fout), C.c_void_p(None),
           C.c_void_p(None), L.ptr(gb))
    ref = 1.0 + go.double().sum(0)
    np.testing.assert_allclose(gb.cpu().numpy(), ref.float().cpu().numpy(), rtol=0, atol=4e-7 * n ** 0.5 * float(ref.abs().max()))
    # and through autograd: a frozen weight with a trainable bias
    x = cu(torch.randn(n, fin, generator=g))
    b = torch.zeros(fout, device=DEV, requires_grad=True)
    models().linear(x, w, b).backward(go)
    np.testing.assert_allclose(b.grad.cpu().numpy(), (ref - 1.0).float().cpu().numpy(), rtol=0, atol=4e-7 * n ** 0.5 * float(ref.abs().max()))


def test_linear_rejects_wide_layers_and_cpu_tensors():
    m = models()
    with pytest.raises(RuntimeError):
        m.linear(torch.zeros(4, 300, device=DEV), torch.zeros(8, 300, device=DEV))
    with pytest.raises(RuntimeError):
        m.linear(torch.zeros(4, 8), torch.zeros(8, 8))


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("in_dim", [96, 256])
def test_f16x2_heads_keep_fp32_accuracy_across_magnitudes(training, in_dim):
    """Round 4: the width-64 heads' forward as two-term fp16 splits (mlp_f2_heads.h).  fp16 has 5 exponent bits, so every operand
    is scaled by a power of two per SAMPLE and layer: rows of one 32-sample tile that differ by 18 orders of magnitude (and rows
    that are exactly zero) must each come out as accurately as the fp32 evaluation of the same row -- per sample, the error against
    an fp64 evaluation may not exceed 4 x torch's fp32 error (plus 2e-6 of the row's own scale)."""
    m = models()
    if m.MATMUL != "f16x2":
        pytest.skip("f16x2 is the default; this run selected another matrix mode")
    torch.manual_seed(7 + in_dim)
    net = m.MLP(in_dim, 64, 3, 3).to(DEV)                 # (the colour decoder's shape: five Linear layers)
    n = 1000
    mag = 10.0 ** (torch.rand(n, 1) * 18.0 - 12.0)
    mag[::7] = 0.0
    mag[1::31] = 1e25                                     # (x 64 inputs x weights: still far below fp32's range)
    x = (torch.randn(n, in_dim) * mag).to(DEV)
    ps = [p.detach().cpu().double() for p in net.params()]

    def chain(v, dtype):
        h = v.to(dtype)
        for l in range(0, len(ps) - 2, 2):
            h = torch.relu(h @ ps[l].to(dtype).T + ps[l + 1].to(dtype))
        return h @ ps[-2].to(dtype).T + ps[-1].to(dtype), h
    ref, h_last = chain(x.cpu(), torch.float64)
    f32, _ = chain(x.cpu(), torch.float32)
    if training:
        y = net.fused(x.clone().requires_grad_(True))     # tn_mlp_fwd_stash
    else:
        with torch.no_grad():
            y = net.fused(x)                              # tn_mlp_fwd
    err = (y.detach().cpu().double() - ref).abs().amax(1)
    err32 = (f32.double() - ref).abs().amax(1)
    scale = (h_last.abs() @ ps[-2].abs().T + ps[-1].abs()).amax(1)          # what one rounding of the last layer is relative to
    assert torch.isfinite(y).all()
    bad = err > 4.0 * err32 + 2e-6 * scale
    assert not bad.any(), (int(bad.sum()), float((err / scale.clamp_min(1e-300)).max()))


@pytest.mark.parametrize("training", [False, True])
def test_f16x2_heads_confine_non_finite_rows(training):
    """fp16 splits turn an infinite operand into NaN (inf - inf in the low term) where the fp32 MFMA would carry the inf -- a difference
    nobody downstream can use, but it must stay in ITS row: scales are per sample, so the 31 other samples of the tile come out
    bit-identical to a run without the poisoned rows.  (What the poisoned rows themselves yield is NOT the reference's NaN: the
    kernels' ReLU is v_max_f32, which returns the other operand for a NaN -- in every matrix mode, fp32 included -- so a NaN dies at
    the first hidden layer where torch.relu would carry it to the output.  Known deviation, non-finite inputs only.)"""
    m = models()
    if m.MATMUL != "f16x2":
        pytest.skip("f16x2 is the default; this run selected another matrix mode")
    torch.manual_seed(11)
    net = m.MLP(96, 64, 3, 3).to(DEV)
    n = 200
    x = torch.randn(n, 96, device=DEV)
    bad = x.clone()
    bad[5, 7] = float("inf")
    bad[37, 0] = float("nan")
    bad[64, 95] = -float("inf")

    def run(inp):
        if training:
            return net.fused(inp.clone().requires_grad_(True)).detach()
        with torch.no_grad():
            return net.fused(inp)
    y, yb = run(x), run(bad)
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[[5, 37, 64]] = False
    assert torch.equal(y[keep], yb[keep])
