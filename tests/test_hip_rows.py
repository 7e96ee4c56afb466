"""Row views between a wide feature stack and the heads behind it (include/tinynerf_hip.h: tn_mlp_desc::x_rows / grad_x_rows,
tn_mlp_rows_view, TN_MLP_GRAD_Y_ROWS) -- the harness plumbing for reference models.py:59-89 on run.py:131-134 (Vanilla,
width 256) and run.py:141-150 (Cobafa, width 128): y^T stays in the stack's workspace as [feature][32-sample] rows, the heads'
first-layer weight gradients read it there (mlp_wgrad_rows.hip) and their data-gradient chains write d loss / d y back in the
same layout.  Oracle: the same kernels through the row-major tensors (the module-API path that the goldens G8 / G11 / G14 pin)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _renderer(method, seed):
    from tinynerf_amd import core, models as m
    torch.manual_seed(seed)
    if method == "vanilla":
        fm, dim = m.VanillaFeatureMLP(10, 256, 8), 256
    else:
        fm = m.CobafaFeatureField(basis_res=[8, 10, 12], coef_res=8, freqs=[2.0, 3.5, 8.0], channels=[8, 8, 4], mlp_hidden_dim=128)
        fm.dropout.p = 0.0
        dim = 128
    r = core.NerfRenderer(fm, m.VanillaOpacityDecoder(dim), m.VanillaColorDecoder(8, dim, 64, 3), torch.ones(3)).to(DEV)
    with torch.no_grad():
        r.sigma_decoder.net.net[2].bias.add_(2.0)
    return r


def _batch(n_rays, per_ray, seed):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(max(1, per_ray // 2), per_ray + 1, (n_rays,), generator=g)
    counts[3] = 0                                                    # an empty ray
    start = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    packed = torch.rand(n, 7, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    packed[:, 3:6] = torch.repeat_interleave(d, counts, dim=0)
    packed[:, 6] = 0.02 + 0.03 * torch.rand(n, generator=g)
    info = torch.stack([start, counts], 1).to(torch.int32)
    return packed.to(DEV), info.to(DEV), torch.rand(n_rays, 3, generator=g).to(DEV)


@pytest.mark.parametrize("method,n_rays,per_ray", [("vanilla", 61, 40), ("vanilla", 700, 57), ("cobafa", 300, 33)])
def test_row_views_equal_row_major_tensors(method, n_rays, per_ray, heads):
    from tinynerf_amd import models as m
    from tinynerf_amd.arena import Arena
    packed, info, target = _batch(n_rays, per_ray, 5)
    assert packed.size(0) % 32 != 0 or method == "cobafa"            # ragged last tile
    grads = {}
    for rows in (False, True):
        r = _renderer(method, 21)
        r.fused = True
        if rows:                                                     # what run.Trainer sets up
            r.reuse_buffers = True
            arena = Arena()
            for i, mod in enumerate(mm for mm in r.feature_module.modules() if isinstance(mm, m.MLP)):
                mod.__dict__["scratch"] = (arena, f"ws{i}", {})
        out = r(packed, info)
        torch.nn.functional.mse_loss(out, target).backward()
        if rows:
            links = [mm.__dict__["scratch"][2] for mm in r.feature_module.modules() if isinstance(mm, m.MLP)]
            assert len(links) == 1 and links[0].get("n") == packed.size(0) and links[0]["delivered"] is False      # consumed by the stack's backward
            assert links[0]["width"] == (256 if method == "vanilla" else 128)
        grads[rows] = {k: p.grad.detach().cpu().numpy() for k, p in r.named_parameters()}
        grads[rows]["__out"] = out.detach().cpu().numpy()
    # the same forward launches up to the colour head's direction encoding (per-ray table vs per-sample sin / cos)
    np.testing.assert_allclose(grads[True]["__out"], grads[False]["__out"], rtol=0, atol=2e-6)
    out_ = grads[True].pop("__out"); grads[False].pop("__out")
    if heads == "fp32":
        # the same products, summed in another order (row tiles / atomics): 2e-5 of the largest element per tensor
        for k, ref in grads[False].items():
            np.testing.assert_allclose(grads[True][k], ref, rtol=0, atol=2e-5 * max(float(np.abs(ref).max()), 1e-30), err_msg=k)
        return
    # f16x2 (default): the row-view path runs the colour head's forward on the fp16 matrix cores (per-ray table, TN_ENC_AUX_CAT) while the
    # row-major path keeps it on the fp32 MFMA (per-sample sin / cos, TN_ENC_DIR_CAT has no f16x2 form) -- two roundings of the same hidden
    # activations, so a unit within an ulp of 0 may fall on either side of its ReLU.  Round 4 allowed a blanket 2e-4 for that; now BOTH
    # paths are held against the CPU port of the reference up to the state of its tie units (tests/_ties.py): 2e-5 of each tensor of the
    # heads and grids, 1e-4 for the ten-layer stack (every layer its own fp32 summation order: the bound of test_hip_parity_r2.py)
    from _ties import assert_grads_match_up_to_relu_ties
    from oracle import torch_port as tp
    r = _renderer(method, 21)
    sd = {k: v.detach().cpu().contiguous() for k, v in r.state_dict().items()}
    pk, inf_, tgt = packed.cpu(), info.cpu(), target.cpu()
    kw = {"vanilla_freqs": 10} if method == "vanilla" else {"cobafa_freqs": (2.0, 3.5, 8.0)}

    def ref():
        return tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, pk, inf_, torch.ones(3), **kw), tgt))[0]
    rel = {k: (1e-4 if k.startswith("feature_module.net") else 2e-5) for k in grads[True]}
    for rows in (True, False):
        assert_grads_match_up_to_relu_ties(grads[rows], ref, rel, weights_conditioning=True, cond_cap=2e-3)


def test_rows_view_reports_only_layer_kernel_stacks():
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    fm = m.VanillaFeatureMLP(10, 256, 8).to(DEV)
    desc = m._mlp_desc(fm.net.params(), 3, L.ENC_POSENC, 10, L.ACT_NONE, fm.encoding.freqs)
    y, g, st = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    assert L.lib().tn_mlp_rows_view(C.byref(desc), C.c_int64(1000), C.byref(y), C.byref(g), C.byref(st)) == 0
    # slab layout (round 5): every row set contiguous over the 32 tiles -- slabs of 32 x 256 rows: 9 hidden activations, buffer A (= y),
    # buffer B (= d loss / d y), and one slab shared by the 64 encoded-input rows and the 9 x 16 ReLU bit rows; a tile of a slab = 256 rows;
    # round 6: + 9 slabs, one per hidden layer's gradient (the cross-layer data-gradient chain writes each once)
    slab = 32 * 256 * 32
    assert y.value == 9 * slab and g.value == 10 * slab and st.value == 256 * 32
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64
    # (+ the tail: per-layer maxima for the f16x2 weight gradient; + round 6: the packed weight stream of the cross-layer forward,
    #  csrc/mlp_fused_f2.hip -- 2.4 MB for this stack whatever n is)
    extra = fn(C.byref(desc), C.c_int64(1000)) - (21 * slab * 4 + 256)
    assert 2 << 20 < extra < 3 << 20 and fn(C.byref(desc), C.c_int64(2016)) - (63 * 21 * 256 * 32 * 4 + 256) == extra
    od = m.VanillaOpacityDecoder(256).to(DEV)                         # a width-64 head has no row views
    d2 = m._mlp_desc(od.net.params(), 256, L.ENC_NONE, 0, L.ACT_EXP_M1, None)
    assert L.lib().tn_mlp_rows_view(C.byref(d2), C.c_int64(1000), C.byref(y), C.byref(g), C.byref(st)) != 0


@pytest.mark.parametrize("n", [1, 777, 40000])
def test_inference_through_the_layer_kernels(n):
    """tn_mlp_fwd_ws (wide stack, layer kernels, two ping-pong row buffers) == the training forward's y bit for bit (same MFMA
    steps in the same order) and == the register-resident inference kernel tn_mlp_fwd to fp32 rounding."""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    torch.manual_seed(3)
    fm = m.VanillaFeatureMLP(10, 256, 8).to(DEV)
    x = (torch.rand(n, 3, device=DEV) * 2 - 1).contiguous()
    with torch.no_grad():
        y_inf = fm(x)                                               # no grad: the workspace form (models._FusedMLP.forward)
    y_train = fm(x.requires_grad_(False))                           # parameters require grad: training forward with stash
    assert y_train.requires_grad
    assert torch.equal(y_inf, y_train.detach())
    desc = m._mlp_desc(fm.net.params(), 3, L.ENC_POSENC, 10, L.ACT_NONE, fm.encoding.freqs)
    y_reg = torch.empty(n, 256, device=DEV)
    L.call("tn_mlp_fwd", x.device, C.byref(desc), L.ptr(x), C.c_void_p(None), C.c_int64(n), L.ptr(y_reg), C.c_void_p(None))
    np.testing.assert_allclose(y_inf.cpu().numpy(), y_reg.cpu().numpy(), rtol=0, atol=2e-6 * float(y_reg.abs().max()))
    fn = L.lib().tn_mlp_fwd_workspace_bytes
    fn.restype = C.c_int64
    rows_bytes = ((n + 31) // 32) * (2 * 256 + 64) * 128          # the layer-wise form's ping-pong rows; behind them the packed weight stream
    assert 2 << 20 < fn(C.byref(desc), C.c_int64(n)) - rows_bytes < 3 << 20      # of the cross-layer form (round 6), 2.4 MB whatever n is
    od = m.VanillaOpacityDecoder(256).to(DEV)
    assert fn(C.byref(m._mlp_desc(od.net.params(), 256, L.ENC_NONE, 0, L.ACT_EXP_M1, None)), C.c_int64(n)) == 0


@pytest.mark.parametrize("split", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("width,n", [(256, 5000), (256, 33), (128, 4099)])
def test_bf16x3_layers_equal_fp32_mfma_layers(width, n, split, monkeypatch):
    """TN_MLP_BF16X3 (mlp_b3_layers.hip: bf16 matrix cores, exact three-way operand splits, six partial products, fp32
    accumulate) and TN_MLP_F16X2 (mlp_f2_layers.hip: two-term fp16 splits with power-of-two scales, the default since round 4)
    against the fp32-MFMA layer kernels on the same wide stack: forward and every gradient to fp32 rounding --
    both are fp32-accurate evaluations of the same sums in different orders (forward 3e-6 of the largest output after ten
    layers, gradients 2e-5 of each tensor's largest element, the bound the fp32 kernels are held to against the oracle)."""
    from tinynerf_amd import models as m
    torch.manual_seed(11)
    net = m.MLP(60 if width == 256 else 36, width, 8 if width == 256 else 5, width).to(DEV)      # Vanilla: PE inputs; Cobafa: plain
    if width == 256:
        fm = m.VanillaFeatureMLP(10, 256, 8).to(DEV)
        fwd = lambda x: fm(x)
        params = list(fm.parameters())
        x = (torch.rand(n, 3, device=DEV) * 2 - 1)
    else:
        fwd = lambda x: net(x)
        params = list(net.parameters())
        x = torch.rand(n, 36, device=DEV)
    g = torch.randn(n, width, device=DEV)
    res = {}
    for mode in ("fp32", split):
        monkeypatch.setattr(m, "MATMUL", mode)
        for p in params:
            p.grad = None
        y = fwd(x)
        y.backward(g)
        with torch.no_grad():
            yi = fwd(x)
        if width == 256:
            assert torch.equal(yi, y.detach())                  # inference (tn_mlp_fwd_ws) == training forward in either mode
        else:                                                   # (plain inputs: inference takes the register-resident fp32 kernel)
            np.testing.assert_allclose(yi.cpu().numpy(), y.detach().cpu().numpy(), rtol=0, atol=3e-6 * float(y.abs().max()))
        res[mode] = (y.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in params])
    y0, g0 = res["fp32"]
    y1, g1 = res[split]
    assert not np.array_equal(y0, y1)                            # (the flag did select another kernel)
    np.testing.assert_allclose(y1, y0, rtol=0, atol=3e-6 * float(np.abs(y0).max()))
    for a_, b_ in zip(g1, g0):
        np.testing.assert_allclose(a_, b_, rtol=0, atol=2e-5 * max(float(np.abs(b_).max()), 1e-30))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("split", ["bf16x3", "f16x2"])
def test_bf16x3_full_size_properties(split, monkeypatch):
    """BASELINE size (2^20 + 17 samples, the ragged last tile included), the Vanilla 256 x 10 stack, everything compared on
    the device:
    * forward, bf16x3 against the fp32-MFMA layer kernels: 3e-6 of the largest output (measured 1.9e-6);
    * backward, linearity in the upstream gradient on the SAME activations (three runs of the deterministic forward): bwd(g1 + g2)
      = bwd(g1) + bwd(g2) to 5e-5 of each tensor's largest element -- sums over 10^6 samples accumulated by atomics in a free
      order repeat to 1e-6 from run to run;
    * backward, bf16x3 against fp32 MFMA: the two forwards differ by 2e-6, so of the 2.4e9 hidden units ~10^3 whose
      pre-activation lies that close to zero take the other ReLU branch, and a sum of 10^8 random-sign terms moves by
      sqrt(10^3 / 10^8) of its norm: 1e-3 measured in the first layers, 3e-6 in the last; bound 5e-3 norm-wise (the tie-aware
      comparisons at test sizes are in tests/_ties.py and the parity tests)."""
    from tinynerf_amd import models as m
    torch.manual_seed(3)
    n = (1 << 20) + 17
    fm = m.VanillaFeatureMLP(10, 256, 8).to(DEV)
    params = list(fm.parameters())
    x = torch.rand(n, 3, device=DEV) * 2 - 1
    # upstream gradients whose rows span more than six orders of magnitude (what volume-rendering weights do to them): f16x2's weight
    # gradient takes ONE power-of-two scale per operand and launch from the launch-wide maxima -- this is the batch that stresses it
    mag = torch.exp(torch.empty(n, 1, device=DEV).uniform_(-16.0, 0.0))
    g1 = torch.randn(n, 256, device=DEV) * 1e-3 * mag
    g2 = torch.randn(n, 256, device=DEV) * 1e-3 * mag.flip(0)

    def run(mode, g):
        monkeypatch.setattr(m, "MATMUL", mode)
        for p in params:
            p.grad = None
        y = fm(x)
        y.backward(g)
        return y.detach(), [p.grad.clone() for p in params]

    y0, ga0 = run("fp32", g1)
    y0 = y0.clone()
    y1, ga = run(split, g1)
    y1 = y1.clone()
    assert torch.isfinite(y1).all() and not torch.equal(y0, y1)
    assert float((y1 - y0).abs().max()) <= 3e-6 * float(y0.abs().max())
    for a_, b_ in zip(ga, ga0):
        assert float((a_ - b_).double().norm()) <= 5e-3 * float(b_.double().norm())
    del y0, ga0
    y2, gb = run(split, g2)
    assert torch.equal(y2, y1)                                   # the forward is deterministic: same activations, same masks
    _, gs = run(split, g1 + g2)
    for a_, b_, s_ in zip(ga, gb, gs):
        assert float((a_ + b_ - s_).abs().max()) <= 5e-5 * max(float(s_.abs().max()), 1e-30)        # (measured 1.2e-5)


@pytest.mark.parametrize("n", [1, 777, 40000])
def test_plain_input_stack_through_the_layer_kernels(n, matmul):
    """Round 4: a wide stack on <= 64 PLAIN inputs -- Cobafa's 36 gathered features into 128 x 6 (models.py:239-247) -- stages x^T
    as 64 zero-padded rows and runs its first layer, that layer's weight gradient and d loss / d x through the row-operand
    kernels (fwd_lds / wgrad_rows<128, 64> / dgrad_first) instead of the general-shape fallbacks.  Inference (tn_mlp_fwd_ws,
    ping-pong scratch) == training forward bit for bit, == the register-resident tn_mlp_fwd to fp32 rounding.  (Gradients of
    these shapes, x included, tie-aware against torch: tests/test_hip_models.py::test_wide_deep_mlp_backward_vs_torch.)"""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    torch.manual_seed(5)
    net = m.MLP(36, 128, 5, 128).to(DEV)
    x = torch.randn(n, 36, device=DEV)
    with torch.no_grad():
        y_inf = net(x)
    xt = x.clone().requires_grad_(True)
    y_train = net(xt)
    assert torch.equal(y_inf, y_train.detach())
    desc = m._mlp_desc(net.params(), 36, L.ENC_NONE, 0, L.ACT_NONE, None)
    fn = L.lib().tn_mlp_fwd_workspace_bytes
    fn.restype = C.c_int64
    assert 0 < fn(C.byref(desc), C.c_int64(n)) - ((n + 31) // 32) * (2 * 128 + 64) * 128 < 1 << 20      # (+ the packed weight stream, 0.4 MB)
    y_reg = torch.empty(n, 128, device=DEV)
    d0 = m._mlp_desc(net.params(), 36, L.ENC_NONE, 0, L.ACT_NONE, None)
    d0.flags = 0                                                   # (fp32 MFMA, register-resident: independent of the layer kernels)
    L.call("tn_mlp_fwd", x.device, C.byref(d0), L.ptr(x), C.c_void_p(None), C.c_int64(n), L.ptr(y_reg), C.c_void_p(None))
    np.testing.assert_allclose(y_inf.cpu().numpy(), y_reg.cpu().numpy(), rtol=0, atol=3e-6 * float(y_reg.abs().max()))


@pytest.mark.parametrize("head,in_dim,n", [("sigma", 256, 777), ("rgb", 256, 2000), ("rgb", 128, 33), ("sigma", 128, 4096)])
def test_head_forward_reads_x_from_row_views(head, in_dim, n):
    """Round 4 ABI: tn_mlp_fwd_stash of a width-64 f16x2 head takes its first-layer operands from tn_mlp_desc::x_rows
    ([feature][32-sample] rows per tile, zeros behind sample n) -- with TN_MLP_X_FROM_ROWS `x` is not dereferenced at all.
    Same values through the same arithmetic: outputs AND the stashed workspace are bit-identical to the row-major call.  The flag
    on a launch that cannot read rows (fp32 heads) is refused with TN_E_CONFIG instead of reading the unwritten x."""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    if m.MATMUL != "f16x2":
        pytest.skip("the row view is read by the f16x2 heads")
    torch.manual_seed(n)
    out = 1 if head == "sigma" else 3
    net = m.MLP(in_dim, 64, 0 if head == "sigma" else 3, out).to(DEV)
    x = torch.randn(n, in_dim, device=DEV) * 3.0
    tiles = (n + 31) // 32
    rows = torch.zeros(tiles, in_dim, 32, device=DEV)
    xp = torch.zeros(tiles * 32, in_dim, device=DEV)
    xp[:n] = x
    rows.copy_(xp.view(tiles, 32, in_dim).transpose(1, 2))
    table = idx = None
    if head == "rgb":                                   # the colour head's per-ray table columns (TN_ENC_AUX_CAT), 56 wide
        table = torch.randn(7, 56, device=DEV)
        table[:, 51:] = 0.0
        idx = torch.randint(0, 7, (n,), dtype=torch.int32, device=DEV)
        net.net[0] = torch.nn.Linear(in_dim + 51, 64).to(DEV)
    ps = net.params()

    def desc(flags):
        if head == "rgb":
            d = m._mlp_desc(ps, in_dim, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, None, flags, idx, 56)
        else:
            d = m._mlp_desc(ps, in_dim, L.ENC_NONE, 0, L.ACT_EXP_M1, None, flags)
        return d
    wsfn = L.lib().tn_mlp_bwd_workspace_bytes
    wsfn.restype = C.c_int64
    nbytes = int(wsfn(C.byref(desc(0)), C.c_int64(n)))
    assert nbytes > 0
    res = {}
    for mode in ("row-major", "rows", "rows-only"):
        d = desc(0)
        xin = x
        if mode != "row-major":
            d.x_rows, d.x_rows_tile_stride = rows.data_ptr(), in_dim * 32
        if mode == "rows-only":
            d.flags |= L.MLP_X_FROM_ROWS
            xin = torch.full_like(x, float("nan"))     # must not be read
        y = torch.empty(n, out, device=DEV)
        ws = torch.zeros(nbytes // 4, device=DEV)
        L.call("tn_mlp_fwd_stash", x.device, C.byref(d), L.ptr(xin), L.ptr(table), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(nbytes))
        res[mode] = (y, ws)
    per_tile = nbytes // 4 // tiles                      # floats per tile: rows x 32 samples
    nh_rows = 64 * (1 if head == "sigma" else 4)        # the stashed hidden activations lead the tile
    n_last = n - 32 * (tiles - 1)

    def stashed(ws):
        t = ws[:tiles * per_tile].view(tiles, per_tile // 32, 32)
        # (the last tile's columns behind sample n belong to nobody: the row-major call evaluates row 0 there, the row view holds zeros)
        return t[:tiles - 1], t[tiles - 1, :nh_rows, :n_last]
    for mode in ("rows", "rows-only"):
        assert torch.equal(res[mode][0], res["row-major"][0]), mode
        for got, want in zip(stashed(res[mode][1]), stashed(res["row-major"][1])):
            assert torch.equal(got, want), mode
    assert torch.isfinite(res["rows-only"][0]).all()
    # a launch that cannot read the rows says so
    d = desc(0)
    d.flags = (d.flags & ~(L.MLP_F16X2 | L.MLP_BF16X3)) | L.MLP_X_FROM_ROWS
    d.x_rows, d.x_rows_tile_stride = rows.data_ptr(), in_dim * 32
    y = torch.empty(n, out, device=DEV)
    ws = torch.zeros(nbytes // 4, device=DEV)
    with pytest.raises(RuntimeError, match="TN_MLP_X_FROM_ROWS"):
        L.call("tn_mlp_fwd_stash", x.device, C.byref(d), L.ptr(x), L.ptr(table), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(nbytes))


def test_slow_general_shape_fallbacks_say_so_once():
    """Round-3 verdict: a caller who lands on one of the general-shape fallback kernels of the wide stacks gets valid results several
    times slower -- with no diagnostic.  Now: once per fallback on stderr and through tn_last_warning_string(); the reference's own
    configurations (a fresh process running one training step of each) never trigger it."""
    import ctypes as C
    import subprocess
    import sys
    from tinynerf_amd import _lib as L, models as m
    L.lib().tn_last_warning_string.restype = C.c_char_p
    net = m.MLP(100, 256, 3, 256).to(DEV)                     # 100 plain inputs: no row-operand first layer for that
    x = torch.randn(500, 100, device=DEV, requires_grad=True)
    net(x).sum().backward()
    msg = L.lib().tn_last_warning_string().decode()
    assert "general-shape" in msg and "256" in msg, msg
    code = ("import torch, ctypes\n"
            "from tinynerf_amd import rays, _lib as L\n"
            "from tinynerf_amd.run import TrainConfig, Trainer\n"
            "o, d, rgb, K, _ = rays.synthetic_scene(n_views=2, res=64, seed=0, device='cuda')\n"
            "for method in ('kplanes', 'vanilla', 'cobafa'):\n"
            "    tr = Trainer(TrainConfig(method=method, scene_type='aabb', batch_size=256, n_samples=32, seed=1, occupancy_res=32), o, d, rgb,\n"
            "                 torch.ones(3, device='cuda'), torch.device('cuda'))\n"
            "    tr.step(); tr.step()\n"
            "    with torch.no_grad():\n"
            "        tr.render_rays(o[:2048], d[:2048])\n"
            "torch.cuda.synchronize()\n"
            "L.lib().tn_last_warning_string.restype = ctypes.c_char_p\n"
            "print('WARNING=[' + L.lib().tn_last_warning_string().decode() + ']')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "WARNING=[]" in r.stdout, (r.stdout[-500:], r.stderr[-500:])


@pytest.mark.parametrize("log2_outlier", [20, 30])
def test_f16x2_weight_gradient_with_an_outlier_sample(log2_outlier, monkeypatch):
    """Round-4 advice: the f16x2 weight gradient of the wide stacks takes ONE power-of-two scale per operand and launch from the launch-wide
    maximum of |G| -- a single outlier sample takes mantissa bits from every other sample of the layer.  One row of the upstream gradient
    2^20 (2^30) times larger than the rest, against an fp64 evaluation of the same stack (oracle/torch_port, double): the f16x2 result may
    be no further from it than 4 x the fp32-MFMA kernels' own distance + 1e-6 of each tensor's largest element.  (What the scale costs the
    small samples -- everything below 2^-34 of the maximum product -- is below what fp32 ACCUMULATION keeps of them next to the outlier's
    term anyway: 2^-24 of the running sum.)"""
    from tinynerf_amd import models as m
    from oracle import torch_port as tp
    torch.manual_seed(17)
    n = 4099
    net = m.MLP(36, 128, 5, 128).to(DEV)                 # Cobafa's stack (run.py:141-147): plain inputs, so that the fp64 reference is exact
    params = list(net.parameters())
    x = torch.rand(n, 36, device=DEV)
    g = torch.randn(n, 128, device=DEV)
    g[n // 3] *= 2.0 ** log2_outlier
    sd64 = {"net." + k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    names = ["net." + k for k, _ in net.named_parameters()]
    leaves = {k: sd64[k].clone().requires_grad_(True) for k in names}
    y64 = tp.mlp({**sd64, **leaves}, "net.net.", x.double().cpu())
    ref = torch.autograd.grad((y64 * g.double().cpu()).sum(), [leaves[k] for k in names])
    err = {}
    for mode in ("fp32", "f16x2"):
        monkeypatch.setattr(m, "MATMUL", mode)
        for p in params:
            p.grad = None
        net(x).backward(g)
        assert all(torch.isfinite(p.grad).all() for p in params)
        err[mode] = [float((p.grad.double().cpu() - r).abs().max() / r.abs().max()) for p, r in zip(params, ref)]
    for e16, e32, name in zip(err["f16x2"], err["fp32"], names):
        assert e16 <= 4 * e32 + 1e-6, (name, e16, e32)


def _bit_rows(x_pos: torch.Tensor, tiles: int, F: int) -> torch.Tensor:
    """the layer kernels' ReLU bit rows of a [tiles * 32, F] boolean matrix: per tile and 32-feature block two 128-byte rows = 64 dwords,
    dword `lane` (sample j = lane & 31, half h = lane >> 5) holds bit r for feature 32 b + (r & 3) + 8 (r >> 2) + 4 h"""
    p = x_pos.view(tiles, 32, F // 32, 32).cpu().numpy()                # [tile][j][block][feature in block]
    out = np.zeros((tiles, F // 32, 64), dtype=np.uint32)
    for r in range(16):
        for h in range(2):
            f = (r & 3) + 8 * (r >> 2) + 4 * h
            out[:, :, 32 * h:32 * h + 32] |= (p[:, :, :, f].transpose(0, 2, 1).astype(np.uint32) << r)
    return torch.from_numpy(out.view(np.int32)).to(DEV)


@pytest.mark.parametrize("in_dim,n,hidden", [(256, 33, False), (256, 2000, True), (128, 777, True), (256, 40037, True), (128, 40037, False)])
def test_heads_pair_backward_over_row_views(in_dim, n, hidden):
    """Round 5: tn_mlp_bwd_pair of the two width-64 heads with x / grad_x as workspace rows -- both chains stop at G_0, grad_x = W_0c[:, x]^T
    G_0c + W_0s^T G_0s is one f16x2 launch from the G_0 rows (heads_dx.hip; `hidden`: x is a ReLU output of the producer and the rows
    leave multiplied by relu'(x) through grad_x_mask_rows), both first layers' x-column weight gradients share one launch
    (tn_mlp_wgrad_rows2).  Oracle: the two single-head tn_mlp_bwd calls (fp32 MFMA, the path G8 / G14 pin) on the same workspaces, and
    fp64 autograd for grad_x."""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    if m.MATMUL != "f16x2":
        pytest.skip("the row views are the f16x2 heads' path")
    torch.manual_seed(7 * n + in_dim)
    sig = m.MLP(in_dim, 64, 0, 1).to(DEV)
    rgb = m.MLP(in_dim + 51, 64, 3, 3).to(DEV)
    x = torch.randn(n, in_dim, device=DEV) * 2.0
    if hidden:
        x = torch.relu(x)
    tiles = (n + 31) // 32
    xp = torch.zeros(tiles * 32, in_dim, device=DEV)
    xp[:n] = x
    rows = xp.view(tiles, 32, in_dim).transpose(1, 2).contiguous()
    bits = _bit_rows(xp > 0, tiles, in_dim) if hidden else None
    table = torch.randn(9, 56, device=DEV)
    table[:, 51:] = 0.0
    idx = torch.randint(0, 9, (n,), dtype=torch.int32, device=DEV)
    mag = torch.exp(torch.empty(n, 1, device=DEV).uniform_(-10.0, 0.0))          # upstream gradients over four orders of magnitude
    g_rgb, g_sig = (torch.randn(n, 3, device=DEV) * mag).contiguous(), (torch.randn(n, 1, device=DEV) * mag).contiguous()
    sp, rp = sig.params(), rgb.params()
    wsfn = L.lib().tn_mlp_bwd_workspace_bytes
    wsfn.restype = C.c_int64

    def descs(flags_r, flags_s, gx_rows):
        rd = m._mlp_desc(rp, in_dim, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, None, flags_r, idx, 56)
        sd = m._mlp_desc(sp, in_dim, L.ENC_NONE, 0, L.ACT_EXP_M1, None, flags_s)
        for d in (rd, sd):
            d.x_rows, d.x_rows_tile_stride = rows.data_ptr(), in_dim * 32
            if gx_rows is not None:
                d.grad_x_rows, d.grad_x_rows_tile_stride = gx_rows.data_ptr(), in_dim * 32
                if bits is not None:
                    d.grad_x_mask_rows, d.grad_x_mask_tile_stride = bits.data_ptr(), (in_dim // 32) * 64
        return rd, sd
    out = {}
    for mode in ("two calls", "pair"):
        rd, sd = descs(L.MLP_X_FROM_ROWS, L.MLP_X_FROM_ROWS, None)
        rb, sb = int(wsfn(C.byref(rd), C.c_int64(n))), int(wsfn(C.byref(sd), C.c_int64(n)))
        ws_r, ws_s = torch.zeros(rb // 4, device=DEV), torch.zeros(sb // 4, device=DEV)
        y_r, y_s = torch.empty(n, 3, device=DEV), torch.empty(n, 1, device=DEV)
        nan_x = torch.full((16,), float("nan"), device=DEV)      # x itself is a placeholder of any size: never dereferenced with row views
        L.call("tn_mlp_fwd_stash", x.device, C.byref(rd), L.ptr(nan_x), L.ptr(table), C.c_int64(n), L.ptr(y_r), L.ptr(ws_r), C.c_int64(rb))
        L.call("tn_mlp_fwd_stash", x.device, C.byref(sd), L.ptr(nan_x), C.c_void_p(None), C.c_int64(n), L.ptr(y_s), L.ptr(ws_s), C.c_int64(sb))
        gx_rows = torch.full((tiles, in_dim, 32), float("nan"), device=DEV)
        g_r, g_s = [torch.zeros_like(p) for p in rp], [torch.zeros_like(p) for p in sp]
        gw_r = (C.c_void_p * 5)(*[g.data_ptr() for g in g_r[0::2]]); gb_r = (C.c_void_p * 5)(*[g.data_ptr() for g in g_r[1::2]])
        gw_s = (C.c_void_p * 2)(*[g.data_ptr() for g in g_s[0::2]]); gb_s = (C.c_void_p * 2)(*[g.data_ptr() for g in g_s[1::2]])
        if mode == "pair":
            rd, sd = descs(L.MLP_STASHED, L.MLP_STASHED, gx_rows)
            L.call("tn_mlp_bwd_pair", x.device, C.byref(rd), C.byref(sd), L.ptr(nan_x), L.ptr(table), L.ptr(g_rgb), L.ptr(g_sig), C.c_int64(n),
                   gw_r, gb_r, gw_s, gb_s, C.c_void_p(None), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
        else:
            rd, sd = descs(L.MLP_STASHED, L.MLP_STASHED | L.MLP_ACCUM_GRAD_X, gx_rows)
            L.call("tn_mlp_bwd", x.device, C.byref(rd), L.ptr(nan_x), L.ptr(table), L.ptr(g_rgb), C.c_int64(n), gw_r, gb_r, C.c_void_p(None),
                   L.ptr(ws_r), C.c_int64(rb))
            L.call("tn_mlp_bwd", x.device, C.byref(sd), L.ptr(nan_x), C.c_void_p(None), L.ptr(g_sig), C.c_int64(n), gw_s, gb_s, C.c_void_p(None),
                   L.ptr(ws_s), C.c_int64(sb))
        gx = gx_rows.transpose(1, 2).reshape(tiles * 32, in_dim)[:n]
        out[mode] = (gx, g_r, g_s, y_r, y_s)
    for a_, b_ in zip(out["pair"][3:], out["two calls"][3:]):
        assert torch.equal(a_, b_)
    gx_p, gx_t = out["pair"][0], out["two calls"][0]
    assert torch.isfinite(gx_p).all()
    scale = float(gx_t.abs().max())
    assert float((gx_p - gx_t).abs().max()) <= 2e-5 * scale
    if hidden:
        assert float(gx_p[x <= 0].abs().max()) == 0.0           # relu'(x) = 0: exactly zero
    for name, (ga, gb_) in {"rgb": (out["pair"][1], out["two calls"][1]), "sigma": (out["pair"][2], out["two calls"][2])}.items():
        for k, (a_, b_) in enumerate(zip(ga, gb_)):
            assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-12), (name, k)
    # fp64: grad_x through the forward's own ReLU masks (the heads' hidden units at a tie are the same on both sides of this comparison)
    xd = x.double().requires_grad_(True)
    dirs_cols = table[idx.long()][:, :51].double()
    z = torch.cat([dirs_cols, xd], 1)
    hr = z
    ps64 = [p.detach().double() for p in rp]
    for l in range(4):
        hr = torch.relu(hr @ ps64[2 * l].t() + ps64[2 * l + 1])
    yr = torch.sigmoid(hr @ ps64[8].t() + ps64[9])
    ss64 = [p.detach().double() for p in sp]
    pre_s = torch.relu(xd @ ss64[0].t() + ss64[1]) @ ss64[2].t() + ss64[3]
    ys = torch.exp(pre_s - 1.0)
    (yr * g_rgb.double()).sum().backward(retain_graph=True)
    gref = xd.grad.clone()
    xd.grad = None
    (ys * g_sig.double()).sum().backward()
    gref = gref + xd.grad
    if hidden:
        gref = gref * (x > 0).double()
    err = float((gx_p.double() - gref).abs().max()) / float(gref.abs().max())
    err_t = float((gx_t.double() - gref).abs().max()) / float(gref.abs().max())
    assert err <= max(1e-4, 2.0 * err_t), (err, err_t)


@pytest.mark.parametrize("which,n", [("vanilla", 1000), ("vanilla", 40037), ("cobafa", 4097)])
def test_stack_without_its_last_layer(which, n):
    """TN_MLP_SKIP_LAST through the C ABI: the stack's training forward stops at its last hidden activation h (bit-identical to the full
    run's: same launches), tn_mlp_rows_view_hidden says where h, the slot for d loss / d (its pre-activation) and its ReLU bit rows are,
    and tn_mlp_bwd started from that slot gives layers 0 .. L - 2 the gradients of the full run, in which the last layer and a linear
    consumer sit behind h (d loss / d y = R: the full run gets R as rows, the short run relu'(h) * (W_last^T R))."""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    if m.MATMUL != "f16x2":
        pytest.skip("TN_MLP_SKIP_LAST rides on the f16x2 layer kernels")
    torch.manual_seed(n)
    if which == "vanilla":
        fm = m.VanillaFeatureMLP(10, 256, 8).to(DEV)
        ps, F, enc, nf, freqs = fm.net.params(), 256, L.ENC_POSENC, 10, fm.encoding.freqs
        x = (torch.rand(n, 3, device=DEV) * 2 - 1).contiguous()
    else:
        net = m.MLP(36, 128, 5).to(DEV)
        ps, F, enc, nf, freqs = net.params(), 128, L.ENC_NONE, 0, None
        x = torch.rand(n, 36, device=DEV).contiguous()
    tiles = (n + 31) // 32
    R = torch.randn(n, F, device=DEV)
    Rp = torch.zeros(tiles * 32, F, device=DEV)
    Rp[:n] = R
    wsfn = L.lib().tn_mlp_bwd_workspace_bytes
    wsfn.restype = C.c_int64
    res = {}
    for skip in (False, True):
        flags = L.MLP_ROWS_ONLY | (L.MLP_SKIP_LAST if skip else 0)
        d = m._mlp_desc(ps, x.size(1), enc, nf, L.ACT_NONE, freqs, flags)
        nb = int(wsfn(C.byref(d), C.c_int64(n)))
        ws = torch.zeros(nb // 4, device=DEV)
        y = torch.full((n, F), float("nan"), device=DEV)
        L.call("tn_mlp_fwd_stash", x.device, C.byref(d), L.ptr(x), C.c_void_p(None), C.c_int64(n), L.ptr(y), L.ptr(ws), C.c_int64(nb))
        assert torch.isnan(y).all()                                   # TN_MLP_ROWS_ONLY: the row-major output stays unwritten
        a_off, g_off, m_off, st = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        if skip:
            L.call_plain("tn_mlp_rows_view_hidden", C.byref(d), C.c_int64(n), C.byref(a_off), C.byref(g_off), C.byref(m_off), C.byref(st))
        else:
            L.call_plain("tn_mlp_rows_view", C.byref(d), C.c_int64(n), C.byref(a_off), C.byref(g_off), C.byref(st))
        assert st.value == F * 32

        def rows(off, count=F):           # `count` rows of every tile, from float offset `off` (a row set may start inside a shared slab)
            return torch.as_strided(ws, (tiles, count, 32), (st.value, 32, 1), off)
        if skip:
            h = rows(a_off.value).transpose(1, 2).reshape(tiles * 32, F)          # the last hidden activation
            assert float(h.min()) >= 0.0
            bits = rows(m_off.value, 2 * (F // 32)).reshape(tiles, F // 32, 64).view(torch.int32)
            assert torch.equal(bits, _bit_rows(h > 0, tiles, F))                   # its ReLU bit rows
            gh = (Rp @ ps[-2].detach()) * (h > 0)                                  # relu'(h) * (W_last^T R)
            rows(g_off.value).copy_(gh.view(tiles, 32, F).transpose(1, 2))
            res["h"] = h
        else:
            rows(g_off.value).copy_(Rp.view(tiles, 32, F).transpose(1, 2))
            res["y"] = rows(a_off.value).transpose(1, 2).reshape(tiles * 32, F)[:n].clone()
        d.flags = L.MLP_STASHED | L.MLP_GRAD_Y_ROWS | (L.MLP_SKIP_LAST if skip else 0) | (d.flags & (L.MLP_F16X2 | L.MLP_BF16X3))
        gs = [torch.zeros_like(p) for p in ps]
        nl = len(ps) // 2
        gw = (C.c_void_p * nl)(*[g.data_ptr() for g in gs[0::2]]); gb = (C.c_void_p * nl)(*[g.data_ptr() for g in gs[1::2]])
        gx = torch.zeros_like(x) if enc == L.ENC_NONE else None
        L.call("tn_mlp_bwd", x.device, C.byref(d), L.ptr(x), C.c_void_p(None), C.c_void_p(None), C.c_int64(n), gw, gb, L.ptr(gx), L.ptr(ws), C.c_int64(nb))
        res[skip] = (gs, gx)
    # forward: y of the full run = W_last h + b_last of the short run's h (to the f16x2 products' 2^-22)
    y_ref = (res["h"][:n] @ ps[-2].t() + ps[-1]).detach()
    assert float((res["y"] - y_ref).abs().max()) <= 2e-5 * float(y_ref.abs().max())
    full, short = res[False][0], res[True][0]
    assert float(short[-2].abs().max()) == 0.0 and float(short[-1].abs().max()) == 0.0      # the last layer's gradients are the caller's business
    for k in range(len(ps) - 2):
        assert float((short[k] - full[k]).abs().max()) <= 3e-5 * max(float(full[k].abs().max()), 1e-12), k
    if res[False][1] is not None:
        assert float((res[True][1] - res[False][1]).abs().max()) <= 3e-5 * float(res[False][1].abs().max())


def test_heads_pair_backward_full_size_properties():
    """The heads' paired backward over row views at the bench's batch size (2^20 + 13 samples behind the 256-wide stack), through
    properties that need no reference: (i) doubling both upstream gradients doubles d loss / d x and every parameter gradient EXACTLY
    (every scale of the f16x2 / bf16x3 products is a power of two and the accumulations are linear), (ii) the first 4 096 samples'
    d loss / d x rows are bit-identical to a 4 096-sample call (a sample's scale is its own; the weights' is the launch's), (iii) rows
    behind sample n stay zero, relu'(x) = 0 entries are exactly zero."""
    import ctypes as C
    from tinynerf_amd import _lib as L, models as m
    if m.MATMUL != "f16x2":
        pytest.skip("the row views are the f16x2 heads' path")
    torch.manual_seed(11)
    in_dim, n = 256, (1 << 20) + 13
    sig = m.MLP(in_dim, 64, 0, 1).to(DEV)
    rgb = m.MLP(in_dim + 51, 64, 3, 3).to(DEV)
    sp, rp = sig.params(), rgb.params()
    tiles = (n + 31) // 32
    rows = torch.relu(torch.randn(tiles, in_dim, 32, device=DEV))
    rows.view(tiles, in_dim, 32)[-1, :, (n - 1) % 32 + 1:] = 0.0                  # samples behind n: zeros, as a producer leaves them
    bits = torch.empty(tiles, in_dim // 32, 64, dtype=torch.int32, device=DEV)
    for t0 in range(0, tiles, 4096):                                               # (bit rows in chunks: the helper works on the host)
        xs = rows[t0:t0 + 4096].transpose(1, 2).reshape(-1, in_dim)
        bits[t0:t0 + 4096] = _bit_rows(xs > 0, xs.size(0) // 32, in_dim)
    table = torch.randn(1024, 56, device=DEV)
    table[:, 51:] = 0.0
    idx = torch.randint(0, 1024, (n,), dtype=torch.int32, device=DEV)
    mag = torch.exp(torch.empty(n, 1, device=DEV).uniform_(-10.0, 0.0))
    g_rgb, g_sig = (torch.randn(n, 3, device=DEV) * mag).contiguous(), (torch.randn(n, 1, device=DEV) * mag).contiguous()
    wsfn = L.lib().tn_mlp_bwd_workspace_bytes
    wsfn.restype = C.c_int64
    dummy = torch.zeros(16, device=DEV)

    def run(nn, scale):
        tl = (nn + 31) // 32
        gx_rows = torch.full((tl, in_dim, 32), float("nan"), device=DEV)

        def descs(flags, with_g):
            rd = m._mlp_desc(rp, in_dim, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, None, flags, idx[:nn].contiguous(), 56)
            sd = m._mlp_desc(sp, in_dim, L.ENC_NONE, 0, L.ACT_EXP_M1, None, flags)
            for d in (rd, sd):
                d.x_rows, d.x_rows_tile_stride = rows.data_ptr(), in_dim * 32
                if with_g:
                    d.grad_x_rows, d.grad_x_rows_tile_stride = gx_rows.data_ptr(), in_dim * 32
                    d.grad_x_mask_rows, d.grad_x_mask_tile_stride = bits.data_ptr(), (in_dim // 32) * 64
            return rd, sd
        rd, sd = descs(L.MLP_X_FROM_ROWS, False)
        rb, sb = int(wsfn(C.byref(rd), C.c_int64(nn))), int(wsfn(C.byref(sd), C.c_int64(nn)))
        ws_r, ws_s = torch.empty(rb // 4, device=DEV), torch.empty(sb // 4, device=DEV)
        y_r, y_s = torch.empty(nn, 3, device=DEV), torch.empty(nn, 1, device=DEV)
        L.call("tn_mlp_fwd_stash", dummy.device, C.byref(rd), L.ptr(dummy), L.ptr(table), C.c_int64(nn), L.ptr(y_r), L.ptr(ws_r), C.c_int64(rb))
        L.call("tn_mlp_fwd_stash", dummy.device, C.byref(sd), L.ptr(dummy), C.c_void_p(None), C.c_int64(nn), L.ptr(y_s), L.ptr(ws_s), C.c_int64(sb))
        g_r, g_s = [torch.zeros_like(p) for p in rp], [torch.zeros_like(p) for p in sp]
        gw_r = (C.c_void_p * 5)(*[g.data_ptr() for g in g_r[0::2]]); gb_r = (C.c_void_p * 5)(*[g.data_ptr() for g in g_r[1::2]])
        gw_s = (C.c_void_p * 2)(*[g.data_ptr() for g in g_s[0::2]]); gb_s = (C.c_void_p * 2)(*[g.data_ptr() for g in g_s[1::2]])
        rd, sd = descs(L.MLP_STASHED, True)
        a_, b_ = (g_rgb[:nn] * scale).contiguous(), (g_sig[:nn] * scale).contiguous()
        L.call("tn_mlp_bwd_pair", dummy.device, C.byref(rd), C.byref(sd), L.ptr(dummy), L.ptr(table), L.ptr(a_), L.ptr(b_), C.c_int64(nn),
               gw_r, gb_r, gw_s, gb_s, C.c_void_p(None), L.ptr(ws_r), C.c_int64(rb), L.ptr(ws_s), C.c_int64(sb))
        torch.cuda.synchronize()
        return gx_rows, g_r + g_s
    gx1, p1 = run(n, 1.0)
    gx2, p2 = run(n, 2.0)
    assert torch.isfinite(gx1).all()
    assert torch.equal(gx2, 2.0 * gx1)                                             # (i) on d loss / d x: exact
    # the weight gradients are sums of atomics: their ORDER differs between two runs -- equal to the order noise of one run against itself
    for a_, b_ in zip(p1, p2):
        assert float((b_ - 2.0 * a_).abs().max()) <= 2e-5 * float(a_.abs().max())
    assert float(gx1[rows <= 0].abs().max()) == 0.0                                # (iii)
    assert float(gx1.view(tiles, in_dim, 32)[-1, :, (n - 1) % 32 + 1:].abs().max()) == 0.0
    gxs, _ = run(4096, 1.0)
    assert torch.equal(gxs, gx1[:128])                                             # (ii)
