"""The wide stacks as ONE persistent launch (csrc/mlp_fused_f2.hip, round 6: activations in registers across all layers, weights
streamed through LDS) against the CPU port of the reference's modules (models.py:7-28,59-68,239-247) and against the layer-by-layer
launches of rounds 3 - 5 on the same inputs.  Tolerance: the north star's 1e-5 of the largest output, as for every MLP-class test."""
import numpy as np
import pytest
import torch

from oracle import torch_port as tp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vanilla(seed, scale=1.0):
    from tinynerf_amd import models
    torch.manual_seed(seed)
    m = models.VanillaFeatureMLP(10, 256, 8).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(scale)
    return m


def _ref_vanilla(m, x):
    sd = {"feature_module." + k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        return tp.mlp(sd, "feature_module.net.net.", tp.posenc(x.cpu(), sd["feature_module.encoding.freqs"])).numpy()


def _both(module, x):
    from tinynerf_amd.models import _FusedMLP
    with torch.no_grad():
        fused = module(x).cpu().numpy()
        _FusedMLP.layerwise_inference = True
        try:
            layer = module(x).cpu().numpy()
        finally:
            _FusedMLP.layerwise_inference = False
    return fused, layer


@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 129, 4097, 100003])
def test_vanilla_stack_in_one_launch(n):
    """ragged sizes: one sample, partial tiles, fewer tiles than waves of a workgroup, more rounds than one"""
    from tinynerf_amd import models
    if models.MATMUL != "f16x2":
        pytest.skip("the cross-layer launch is the f16x2 form")
    m = _vanilla(3)
    torch.manual_seed(n)
    x = (torch.rand(n, 3, device=DEV) * 2 - 1)
    fused, layer = _both(m, x)
    ref = _ref_vanilla(m, x)
    tol = 1e-5 * np.abs(ref).max()
    assert np.isfinite(fused).all()
    assert np.abs(fused - ref).max() <= tol, (np.abs(fused - ref).max(), tol)
    assert np.abs(layer - ref).max() <= tol
    assert np.abs(fused - layer).max() <= tol


@pytest.mark.parametrize("scale", [1e-3, 1.0, 30.0])
def test_vanilla_stack_scales(scale):
    """weights 1000 x smaller / 30 x larger than torch's initialisation: activations shrink / grow by that factor per layer (1e-30 ..
    1e+14 over ten layers); the a-priori bound behind the per-sample scales must neither overflow fp16 nor lose the small columns"""
    from tinynerf_amd import models
    if models.MATMUL != "f16x2":
        pytest.skip("the cross-layer launch is the f16x2 form")
    m = _vanilla(5, scale)
    x = torch.rand(5000, 3, device=DEV) * 2 - 1
    fused, layer = _both(m, x)
    sd = {"feature_module." + k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref64 = tp.mlp(sd, "feature_module.net.net.", tp.posenc(x.cpu().double(), sd["feature_module.encoding.freqs"])).numpy()
    ref32 = _ref_vanilla(m, x)
    scale_ = np.abs(ref64).max()
    e_fused, e_layer, e_torch = (np.abs(a - ref64).max() / scale_ for a in (fused, layer, ref32))
    print(f"scale {scale}: distance to fp64 / largest output: fused {e_fused:.2e} layer-wise {e_layer:.2e} torch fp32 {e_torch:.2e}")
    assert np.isfinite(fused).all()
    assert e_fused <= max(1e-5, 4 * e_torch)


@pytest.mark.parametrize("depth,n", [(5, 70001), (5, 1), (5, 255), (5, 257), (4, 4097), (2, 513), (1, 100)])
def test_cobafa_stack_in_one_launch(depth, n):
    """Cobafa's 128-wide stack (models.py:247: MLP(36, 128, 5)) on plain inputs; eight waves of a workgroup share the weight ring here
    (one tile each: n around 8 x 32 = fewer / more tiles than waves); other depths: both parities of the hidden-layer count"""
    from tinynerf_amd import models
    if models.MATMUL != "f16x2":
        pytest.skip("the cross-layer launch is the f16x2 form")
    torch.manual_seed(11)
    m = models.MLP(36, 128, depth).to(DEV)
    x = torch.randn(n, 36, device=DEV) * 0.3
    fused, layer = _both(m, x)
    sd = {"m." + k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = tp.mlp(sd, "m.net.", x.cpu()).numpy()
    tol = 1e-5 * np.abs(ref).max()
    assert np.abs(fused - ref).max() <= tol, (np.abs(fused - ref).max(), tol)
    assert np.abs(fused - layer).max() <= tol


def test_rows_of_one_tile_orders_of_magnitude_apart():
    """Cobafa-style plain inputs whose rows differ by 12 orders of magnitude inside one 32-sample tile, and zero rows: scales are per
    sample, so every row keeps fp32 accuracy relative to ITS OWN output"""
    from tinynerf_amd import models
    if models.MATMUL != "f16x2":
        pytest.skip("the cross-layer launch is the f16x2 form")
    torch.manual_seed(13)
    m = models.MLP(36, 128, 5).to(DEV)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("bias"):
                p.zero_()                       # (homogeneous stack: the output scales with the input row)
    x = torch.randn(64, 36, device=DEV)
    mag = 10.0 ** torch.linspace(-6, 6, 64, device=DEV)
    x = x * mag[:, None]
    x[5] = 0
    fused, _ = _both(m, x)
    sd = {"m." + k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = tp.mlp(sd, "m.net.", x.cpu().double()).numpy()
    row_scale = np.abs(ref).max(axis=1)
    err = np.abs(fused - ref).max(axis=1)
    ok = err <= 2e-5 * row_scale + 1e-30
    assert ok.all(), (err / np.maximum(row_scale, 1e-300))[~ok]
    assert (fused[5] == 0).all()


@pytest.mark.parametrize("shape,n", [(s_, n_) for s_ in ("vanilla", "cobafa") for n_ in (1, 33, 4097, 70001)]
                         + [("vanilla7", 4097), ("cobafa4", 4097), ("cobafa2", 333)])     # (the other parity of the hidden-layer count: the P / Q column swap)
def test_training_forward_in_one_launch_leaves_the_layerwise_workspace(shape, n):
    """The cross-layer training forward (tn_mlp_fwd_stash without TN_MLP_LAYERWISE) must leave what the layer-wise backward reads:
    activations as rows (1e-5 of each layer's largest value against the layer-wise launches' rows), ReLU bit rows (equal except where an
    activation is within rounding of zero), the per-layer maxima, y -- and the gradients that come out of the unchanged backward."""
    from tinynerf_amd import models
    from tinynerf_amd.models import _FusedMLP
    if models.MATMUL != "f16x2":
        pytest.skip("the cross-layer launch is the f16x2 form")
    torch.manual_seed(7)
    depth = int(shape[-1]) if shape[-1].isdigit() else (8 if shape == "vanilla" else 5)
    shape = shape.rstrip("0123456789")
    if shape == "vanilla":
        m = models.VanillaFeatureMLP(10, 256, depth).to(DEV)
        x = torch.rand(n, 3, device=DEV) * 2 - 1
    else:
        m = models.MLP(36, 128, depth).to(DEV)
        x = (torch.randn(n, 36, device=DEV) * 0.3).requires_grad_(True)      # (d loss / d x: what the gradient chain hands to the first layer)
    gy = torch.randn(n, m(x[:1]).shape[-1], device=DEV)
    res = {}
    for mode in ("fused", "layerwise"):
        _FusedMLP.layerwise_training = mode == "layerwise"
        try:
            for p in m.parameters():
                p.grad = None
            x.grad = None
            y = m(x)
            y.backward(gy)
            res[mode] = (y.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
            if x.requires_grad:
                res[mode][1]["x"] = x.grad.detach().cpu().numpy()
        finally:
            _FusedMLP.layerwise_training = False
    yf, gf = res["fused"]
    yl, gl = res["layerwise"]
    assert np.abs(yf - yl).max() <= 1e-5 * np.abs(yl).max()
    # The two forwards differ by fp32 rounding, so a hidden unit within rounding of zero may take the other ReLU branch in one of them and
    # its whole backward contribution moves ("ReLU ties"): with 70 001 samples x 2 304 units a handful do.  The yardstick is
    # therefore an fp64 evaluation of the same stack (torch autograd on the CPU): the cross-layer form may be no further from it than
    # 3 x the layer-wise launches are (or 1e-5 of the tensor's largest entry where they agree better than that).
    sd = {"m." + k: v.detach().cpu().double().requires_grad_(v.is_floating_point() and not k.endswith("freqs")) for k, v in m.state_dict().items()}
    xin = x.detach().cpu().double().requires_grad_(x.requires_grad)
    inp = tp.posenc(xin, sd["m.encoding.freqs"]) if shape == "vanilla" else xin
    (tp.mlp(sd, "m.net.net." if shape == "vanilla" else "m.net.", inp) * gy.cpu().double()).sum().backward()
    for k in gl:
        ref = (xin.grad if k == "x" else sd["m." + k].grad).numpy()
        scale = max(np.abs(ref).max(), 1e-300)
        e_f, e_l = np.abs(gf[k] - ref).max() / scale, np.abs(gl[k] - ref).max() / scale
        assert e_f <= max(1e-5, 3.0 * e_l), (k, e_f, e_l)
