"""GPU: TN_MLP_LEAN (round 5) -- the paired K-Planes heads without a stash of hidden activations.

The training forward writes ReLU masks, the last pre-activation and the feature rows; the weight-gradient half of the backward rebuilds
H_1 .. H_4 from the feature rows with the forward's f16x2 arithmetic in both MFMA orientations and multiplies them with the chain's G
rows as exact bf16 triplets (csrc/mlp_wgrad_rc.hip).  Reference: the autograd of src/models.py:7-28,70-89 under src/run.py:259.

* C ABI, tn_mlp_fwd_stash_pair / tn_mlp_bwd_pair: lean against the stash form (outputs and d / dx the same bits: the chain is the same
  launch) and BOTH against an fp64 autograd evaluation of the same modules, ragged sizes; the workspace's H rows stay untouched.
* the fused K-Planes renderer at BASELINE config 3's size (2^20 + 13 samples, 128 / 256 / 512 planes) in both head forms: rendered
  colours of a ray subset against the CPU port, every gradient additive over a split of the batch.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _heads(F=96):
    from tinynerf_amd import models as m
    sig = m.VanillaOpacityDecoder(F).to(DEV)
    cd = m.VanillaColorDecoder(8, F, 64, 3).to(DEV)
    return sig, cd


def _pair_run(sig, cd, x, ray_ids, table, g_rgb, g_sig, lean, poison=False):
    from tinynerf_amd import _lib as L
    from tinynerf_amd.models import _mlp_desc
    dev = x.device
    n, F = x.shape
    sp = [p.detach().contiguous() for p in sig.net.params()]
    rp = [p.detach().contiguous() for p in cd.net.params()]
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64
    flags = L.MLP_LEAN if lean else 0
    rd = _mlp_desc(rp, F, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs, flags, ray_ids, 56)
    sd = _mlp_desc(sp, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, flags, None, 0)
    if lean:
        assert L.lib().tn_mlp_lean_supported(C.byref(rd), C.byref(sd)) == 1
    nbr, nbs = int(fn(C.byref(rd), C.c_int64(n))), int(fn(C.byref(sd), C.c_int64(n)))
    fill = float("nan") if poison else 0.0
    wr, wsg = torch.full((nbr // 4,), fill, device=dev), torch.full((nbs // 4,), fill, device=dev)
    rgb, sigma = torch.empty(n, 3, device=dev), torch.empty(n, 1, device=dev)
    L.call("tn_mlp_fwd_stash_pair", dev, C.byref(rd), C.byref(sd), L.ptr(x), L.ptr(table), C.c_int64(n), L.ptr(rgb), L.ptr(sigma),
           L.ptr(wr), C.c_int64(nbr), L.ptr(wsg), C.c_int64(nbs))
    rd.flags |= L.MLP_STASHED
    sd.flags |= L.MLP_STASHED
    grs, gss = [torch.zeros_like(p) for p in rp], [torch.zeros_like(p) for p in sp]
    arr = lambda gs, o: (C.c_void_p * (len(gs) // 2))(*[g.data_ptr() for g in gs[o::2]])
    gx = torch.empty(n, F, device=dev)
    L.call("tn_mlp_bwd_pair", dev, C.byref(rd), C.byref(sd), L.ptr(x), L.ptr(table), L.ptr(g_rgb), L.ptr(g_sig), C.c_int64(n),
           arr(grs, 0), arr(grs, 1), arr(gss, 0), arr(gss, 1), L.ptr(gx), L.ptr(wr), C.c_int64(nbr), L.ptr(wsg), C.c_int64(nbs))
    return rgb, sigma, grs + gss, gx, (wr, wsg)


def _fp64_grads(sig, cd, x, dirs, g_rgb, g_sig):
    """torch autograd in fp64 of models.py:70-89 on the same inputs, upstream gradients g_rgb / g_sig"""
    import copy
    s64, c64 = copy.deepcopy(sig).double(), copy.deepcopy(cd).double()
    xd, dd = x.double(), dirs.double()
    fr = c64.pe.freqs
    pe = torch.cat([torch.sin(dd[..., None] * fr), torch.cos(dd[..., None] * fr)], -1).flatten(-2)
    rgb = torch.sigmoid(c64.net.net(torch.cat([pe, dd, xd], -1)))
    sg = torch.exp(s64.net.net(xd) - 1.0)
    ps = list(c64.net.params()) + list(s64.net.params())
    return torch.autograd.grad((rgb * g_rgb.double()).sum() + (sg * g_sig.double()).sum(), ps)


@pytest.mark.parametrize("n", [1, 31, 33, 999, 40037])
def test_lean_pair_equals_the_stash_form_and_fp64(n, monkeypatch):
    from tinynerf_amd import _lib as L, models
    monkeypatch.setattr(models, "MATMUL", "f16x2")
    torch.manual_seed(100 + n)
    sig, cd = _heads()
    dev = torch.device(DEV)
    R = max(1, n // 40)
    x = torch.rand(n, 96, device=dev) * torch.rand(n, 1, device=dev)          # rows of different magnitude inside a tile
    ray_ids = torch.sort(torch.randint(0, R, (n,), device=dev, dtype=torch.int32)).values.contiguous()
    dirs_ray = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    table = torch.empty(R, 56, device=dev)
    L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(cd.pe.freqs), C.c_int(8), L.ptr(table), C.c_int(56))
    # upstream gradients over six orders of magnitude (volume-rendering weights do that): the weight gradient's operands are exact
    # bf16 triplets, no scale is shared between samples
    mag = torch.exp(torch.empty(n, 1, device=dev).uniform_(-14.0, 0.0))
    g_rgb, g_sig = torch.randn(n, 3, device=dev) * mag, torch.randn(n, 1, device=dev) * mag
    rgb0, s0, g0, gx0, _ = _pair_run(sig, cd, x, ray_ids, table, g_rgb, g_sig, lean=False)
    rgb1, s1, g1, gx1, (wr, wsg) = _pair_run(sig, cd, x, ray_ids, table, g_rgb, g_sig, lean=True, poison=True)
    assert torch.equal(rgb0, rgb1) and torch.equal(s0, s1) and torch.equal(gx0, gx1)          # same forward arithmetic, same chain launch
    # the lean forward left every H row alone (and nothing read them: the gradients below are finite)
    tiles = (n + 31) // 32
    rows_r, rows_s = wr.view(tiles, -1, 32), wsg.view(tiles, -1, 32)
    assert torch.isnan(rows_r[:, :4 * 64]).all() and torch.isnan(rows_s[:, :64]).all()
    assert not torch.isnan(rows_r[:, 4 * 64:8 * 64 + 4]).any()               # G rows and g_pre: written by the chain
    ref = _fp64_grads(sig, cd, x, dirs_ray[ray_ids.long()], g_rgb, g_sig)
    for a_, b_, r_ in zip(g1, g0, ref):
        assert torch.isfinite(a_).all()
        scale = float(r_.abs().max())
        e_lean, e_stash = float((a_.double() - r_).abs().max()) / scale, float((b_.double() - r_).abs().max()) / scale
        # both forms are fp32 evaluations of the same sums: they agree to 2e-5 of the tensor's largest element, and the lean form is no
        # further from fp64 than 1.5 x the stash form + 4e-6 (the order of the flush atomics alone moves either by ~1e-6).  (Their common distance from fp64 can be larger: a hidden unit whose
        # pre-activation is a rounding away from 0 takes the other ReLU branch in fp64, and with upstream gradients over six orders
        # of magnitude one sample can carry a visible share of a sum -- tests/_ties.py; both forms share the forward's masks.)
        e_pair = float((a_ - b_).abs().max()) / scale
        assert e_pair <= 2e-5 and e_lean <= 1.5 * e_stash + 4e-6, (tuple(a_.shape), e_pair, e_lean, e_stash)


def _kplanes_renderer(seed, res=(128, 256, 512)):
    from tinynerf_amd import core, models as m
    torch.manual_seed(seed)
    field = m.KPlanesFeatureField(32, res)
    return core.NerfRenderer(field, m.VanillaOpacityDecoder(96), m.VanillaColorDecoder(8, 96, 64, 3), torch.ones(3)).to(DEV)


def _ragged_batch(n, n_rays, seed, dev=DEV):
    g = torch.Generator().manual_seed(seed)
    cuts = torch.sort(torch.randint(0, n + 1, (n_rays - 1,), generator=g)).values
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([n])])
    cnt = (bounds[1:] - bounds[:-1]).to(torch.int32)
    info = torch.stack([torch.cumsum(cnt, 0, dtype=torch.int32) - cnt, cnt], -1).to(dev)
    packed = torch.rand(n, 7, generator=g).to(dev)
    packed[:, :3] = packed[:, :3] * 2.0 - 1.0
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1).to(dev)
    packed[:, 3:6] = d[torch.repeat_interleave(torch.arange(n_rays, device=dev), cnt.to(dev).long())]
    packed[:, 6] = 0.004
    return packed, info


@pytest.mark.parametrize("lean", [True, False])
def test_fused_render_lean_small(lean, monkeypatch):
    """the fused K-Planes node with and without TN_MLP_LEAN on a ragged batch, NaN-poisoned workspaces: same colours (bits), gradients
    equal to 2e-5 of each tensor; the launches say which form ran"""
    from tinynerf_amd import fused, models
    monkeypatch.setattr(models, "MATMUL", "f16x2")
    r = _kplanes_renderer(3, (16, 40, 96))
    packed, info = _ragged_batch(7013, 211, 4)
    target = torch.rand(info.size(0), 3, device=DEV)
    res = {}
    orig_ws = fused._workspace

    def poisoned(*a, **k):
        t, nb = orig_ws(*a, **k)
        if t is not None:
            t.fill_(float("nan"))
        return t, nb
    monkeypatch.setattr(fused, "_workspace", poisoned)
    for form in (False, True):
        monkeypatch.setattr(fused, "KP_LEAN", form)
        r.zero_grad(set_to_none=True)
        out = r(packed, info)
        torch.nn.functional.mse_loss(out, target).backward()
        res[form] = (out.detach().clone(), {k: p.grad.clone() for k, p in r.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    for k, g in res[True][1].items():
        assert torch.isfinite(g).all(), k
        np.testing.assert_allclose(g.cpu().numpy(), res[False][1][k].cpu().numpy(), rtol=0, atol=2e-5 * float(g.abs().max()), err_msg=k)


def test_fused_pair_full_size_properties(heads):
    """tn_kplanes_mlp_fwd_pair / tn_kplanes_mlp_bwd_pair at BASELINE config 3's size -- 2^20 + 13 packed samples, 128 / 256 / 512 planes:
    32-bit byte offsets into the 32 MiB planes, > 32 k tiles per launch -- in both head forms (f16x2: the lean form; fp32: the stash):
    (i) rendered colours of a subset of rays against the CPU port of the reference (core.py:225-267); (ii) every parameter gradient --
    nine planes, both heads -- additive over a split of the batch's rays: bwd(all) == bwd(first part) + bwd(rest)."""
    from oracle import torch_port as tp
    r = _kplanes_renderer(11)
    n, R = (1 << 20) + 13, 22571
    packed, info = _ragged_batch(n, R, 12)
    with torch.no_grad():
        r.sigma_decoder.net.net[2].bias += 3.0            # a medium in which weights span orders of magnitude and some rays terminate
    up = torch.randn(R, 3, device=DEV)

    def run(lo, hi):                                      # rays [lo, hi)
        s0 = int(info[lo, 0])
        s1 = int(info[hi - 1, 0] + info[hi - 1, 1])
        inf = info[lo:hi].clone()
        inf[:, 0] -= s0
        r.zero_grad(set_to_none=True)
        out = r(packed[s0:s1].contiguous(), inf.contiguous())
        out.backward(up[lo:hi])
        return out.detach(), {k: p.grad.clone() for k, p in r.named_parameters()}
    out, g_all = run(0, R)
    # (i) a subset of rays through the CPU port
    sd = {k: v.detach().cpu().contiguous() for k, v in r.state_dict().items()}
    pick = [0, 1, 2, R // 3, R // 2, R - 2, R - 1] + list(range(1000, 1040))
    for q in pick:
        s0, c = int(info[q, 0]), int(info[q, 1])
        ref = tp.render(sd, packed[s0:s0 + c].cpu(), torch.tensor([[0, c]], dtype=torch.int32), torch.ones(3))
        np.testing.assert_allclose(out[q].cpu().numpy(), ref[0].detach().numpy(), rtol=0, atol=1e-5, err_msg=f"ray {q}")
    # (ii) additivity over a split that is not tile-aligned
    cut = 8191
    _, g_a = run(0, cut)
    _, g_b = run(cut, R)
    for k in g_all:
        s = (g_a[k] + g_b[k]).cpu().numpy()
        np.testing.assert_allclose(g_all[k].cpu().numpy(), s, rtol=2e-3, atol=2e-4 * max(float(np.abs(s).max()), 1e-30), err_msg=k)
        assert np.isfinite(s).all() and float(np.abs(s).max()) > 0, k
