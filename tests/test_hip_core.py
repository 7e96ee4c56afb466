"""GPU parity tests of the renderer core (tinynerf_amd.core -> libtinynerf_hip.so) against the CPU
oracle and the golden vectors captured from the reference.  Integer / index work and sampled
coordinates: bit-exact.  Floating point: 1e-5 (north star), tolerance stated per assert."""
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


def bits(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def core():
    from tinynerf_amd import core as c
    return c


def ragged(rng, n_rays, max_len, p_empty=0.1):
    cnt = rng.integers(1, max_len + 1, n_rays).astype(np.int32)
    cnt[rng.random(n_rays) < p_empty] = 0
    start = (np.cumsum(cnt) - cnt).astype(np.int32)
    return np.stack([start, cnt], -1), int(cnt.sum())


def danger_rays(s, d, info, thr, rel=1e-4):
    """rays whose transmittance passes within `rel` of the threshold: the termination index there
    depends on the last ulp of exp(), which no two libm/GPU implementations share."""
    bad = np.zeros(info.shape[0], bool)
    for r, (a, c) in enumerate(info):
        T = np.concatenate([[1.0], np.cumprod(np.exp(-s[a:a + c].astype(np.float64) * d[a:a + c]))])
        bad[r] = (np.abs(T - thr) < rel * thr).any()
    return bad


# ------------------------------------------------------------------ weights (a17)
@pytest.mark.parametrize("max_len,n_rays", [(40, 300), (1024, 64), (3000, 5)])
def test_weights_fwd_bwd_vs_oracle(max_len, n_rays):
    rng = np.random.default_rng(max_len)
    info, n = ragged(rng, n_rays, max_len)
    s = (rng.random(n) * (60.0 if max_len > 100 else 30.0)).astype(np.float32)
    d = (rng.random(n) * (0.01 if max_len > 100 else 0.08) + 0.001).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    w_ref = orc.weights_fwd(s, d, info, 1e-4)
    st = cu(s).requires_grad_(True)
    w = core().NerfWeights.apply(st, cu(d), cu(info, torch.int32), 1e-4)
    bad = danger_rays(s, d, info, 1e-4)
    # a ray is excluded only when its transmittance passes within 1e-4 (relative) of the threshold: with T falling by a
    # factor >= 1.001 per sample that is at most a handful of rays per case -- bounded, so the exclusion cannot hide a bug
    assert bad.sum() <= max(2, n_rays // 50), int(bad.sum())
    ok = ~np.repeat(bad, info[:, 1])
    np.testing.assert_allclose(w.detach().cpu().numpy()[ok], w_ref[ok], rtol=0, atol=1e-6)
    assert (w_ref == 0).any() and np.array_equal((w.detach().cpu().numpy() == 0)[ok], (w_ref == 0)[ok])
    w.backward(cu(g))
    gs_ref = orc.weights_bwd(s, d, info, w_ref, g)
    np.testing.assert_allclose(st.grad.cpu().numpy()[ok], gs_ref[ok], rtol=1e-4, atol=TOL)


def test_weights_edge_cases():
    c = core()
    # empty ray list, all-empty rays, single-sample ray
    w = c.NerfWeights.apply(torch.rand(0, device=DEV), torch.rand(0, device=DEV), torch.zeros((3, 2), dtype=torch.int32, device=DEV), 1e-4)
    assert w.numel() == 0
    info = torch.tensor([[0, 1], [1, 0], [1, 2]], dtype=torch.int32, device=DEV)
    s = torch.tensor([2.0, 3.0, 1.0], device=DEV); d = torch.tensor([0.5, 0.25, 0.1], device=DEV)
    w = c.NerfWeights.apply(s, d, info, 1e-4).cpu().numpy()
    np.testing.assert_allclose(w[0], 1 - np.exp(-1.0), rtol=1e-6)
    np.testing.assert_allclose(w[2], np.exp(-0.75) * (1 - np.exp(-0.1)), rtol=1e-5)
    with pytest.raises(RuntimeError):
        c.NerfWeights.apply(s, d, info.long(), 1e-4)          # info must be int32 (cuda.cu:89)


def test_weights_full_size_properties():
    """BASELINE size (N = 2^20): sum_k w_k = 1 - T_end and w >= 0, checked without the oracle."""
    rng = np.random.default_rng(0)
    info, n = ragged(rng, 8192, 256, 0.05)
    s = torch.rand(n, device=DEV) * 20; d = torch.full((n,), 0.005, device=DEV)
    it = cu(info, torch.int32)
    w = core().NerfWeights.apply(s, d, it, 0.0)
    seg = torch.repeat_interleave(torch.arange(info.shape[0], device=DEV), it[:, 1].long())
    opac = torch.zeros(info.shape[0], device=DEV, dtype=torch.float64).index_add_(0, seg, w.double())
    logT = torch.zeros(info.shape[0], device=DEV, dtype=torch.float64).index_add_(0, seg, (-s * d).double())
    np.testing.assert_allclose(opac.cpu().numpy(), (1 - torch.exp(logT)).cpu().numpy(), atol=2e-5)
    assert (w >= 0).all()


# ------------------------------------------------------------------ composite (a18)
def test_composite_fwd_bwd_vs_oracle():
    rng = np.random.default_rng(7)
    info, n = ragged(rng, 200, 90)
    rgb = rng.random((n, 3)).astype(np.float32); w = (rng.random(n) * 0.05).astype(np.float32)
    w[rng.random(n) < 0.3] = 0
    bg = np.array([1.0, 0.5, 0.25], np.float32)
    from tinynerf_amd.core import _Composite
    for b in (bg, None):
        rt, wt = cu(rgb).requires_grad_(True), cu(w).requires_grad_(True)
        out = _Composite.apply(rt, wt, cu(info, torch.int32), None if b is None else cu(b))
        np.testing.assert_allclose(out.detach().cpu().numpy(), orc.composite(rgb * (w[:, None] != 0), w, info, b), atol=TOL)
        go = rng.standard_normal((info.shape[0], 3)).astype(np.float32)
        out.backward(cu(go))
        seg = np.repeat(np.arange(info.shape[0]), info[:, 1])
        np.testing.assert_allclose(rt.grad.cpu().numpy(), w[:, None] * go[seg], atol=TOL)
        gw = ((rgb * (w[:, None] != 0)) * go[seg]).sum(-1) - (0 if b is None else (b[None] * go[seg]).sum(-1))
        np.testing.assert_allclose(wt.grad.cpu().numpy(), gw, atol=TOL)


# ------------------------------------------------------------------ marchers / contractions (a1-a4)
def test_march_aabb_bit_exact():
    c = core()
    for name in ("G1_march_aabb", "G1b_march_aabb_asym"):
        g = load_golden(name)
        m = c.RayMarcherAABB(cu(g["aabb"]), int(g["n_samples"]), float(g["near"]), float(g["far"]))
        assert bits(m.step_size) == bits(g["step_size"])
        t, dl = m(cu(g["rays_o"]), cu(g["rays_d"]))
        assert np.array_equal(bits(t), bits(g["t"])) and np.array_equal(bits(dl), bits(g["delta"]))
        if "coords" in g:
            pts = cu(g["rays_o"])[:, None, :] + cu(g["rays_d"])[:, None, :] * t[..., None]
            cc, mask = c.ContractionAABB(cu(g["aabb"]))(pts)
            assert np.array_equal(mask.cpu().numpy(), g["mask"])
            assert np.array_equal(bits(cc), bits(g["coords"]))


@pytest.mark.parametrize("S", [8, 200, 1000])
def test_march_unbounded(S):
    c = core()
    g = load_golden(f"G2_unbounded_S{S}")
    m = c.RayMarcherUnbounded(S, float(g["near"]), 1e5, float(g["uniform_range"]))
    t, dl = m(cu(g["rays_o"]), cu(g["rays_d"]))
    assert t.shape == (g["rays_o"].shape[0], S)
    # torch.linspace on the device rounds differently from the CPU kernel (both are "torch"): 1-ulp class
    np.testing.assert_allclose(t[0].cpu().numpy(), g["t_row"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(dl[0].cpu().numpy(), g["delta_row"], rtol=2e-3, atol=2e-6)
    # contraction itself is bit-exact when fed identical points
    tt = cu(g["t_row"])
    pts = cu(g["rays_o"])[:, None, :] + cu(g["rays_d"])[:, None, :] * tt[None, :, None]
    cc, mask = c.ContractionMip360(float("inf"))(pts)
    assert mask is None and np.array_equal(bits(cc), bits(g["coords_inf"]))
    cc2, _ = c.ContractionMip360(2)(pts)
    np.testing.assert_allclose(cc2.cpu().numpy(), g["coords_l2"], atol=2e-7)


# ------------------------------------------------------------------ occupancy (a5, a6)
def test_occupancy_query_bit_exact():
    c = core()
    g = load_golden("G3_occupancy_query")
    og = c.OccupancyGrid(list(g["grid"].shape), 1 / 1024.).to(DEV)
    og.grid.copy_(cu(g["grid"]))
    og.mean = float(g["grid"].mean())
    assert og.threshold == pytest.approx(float(g["threshold"]))
    assert np.array_equal(og(cu(g["coords"])).cpu().numpy(), g["occupied"])
    # raw interpolated values through the C ABI: bit-exact vs ATen's CPU grid_sampler_3d
    import ctypes as C
    from tinynerf_amd import _lib as L
    vals = torch.empty(g["coords"].shape[0], device=DEV)
    D, H, W = g["grid"].shape
    L.call("tn_occupancy_query", vals.device, L.ptr(og.grid), C.c_int(D), C.c_int(H), C.c_int(W), L.ptr(cu(g["coords"])),
           C.c_int64(vals.numel()), C.c_float(0.01), C.c_void_p(None), L.ptr(vals))
    assert np.array_equal(bits(vals), bits(g["values"]))


def test_occupancy_reference_known_answer():
    """reference tests/test_core.py:5-38."""
    c = core()
    g = load_golden("G3b_reference_known_answer")
    og = c.OccupancyGrid(128, 1 / 1024.).to(DEV)
    og.grid[:, :, 64:] = 0.
    assert og.grid.sum().item() >= og.grid.numel() / 3. and og.grid.sum().item() <= 2. * og.grid.numel() / 3.
    assert np.array_equal(og(cu(g["coords"])).cpu().numpy(), g["occupied"])
    assert og.occupancy() == pytest.approx(0.5)


def test_occupancy_update_vs_golden():
    c = core()
    from tinynerf_amd import models
    g = load_golden("G5_occupancy_update")
    fm = models.VanillaFeatureMLP(4, 32, 2); od = models.VanillaOpacityDecoder(32)
    fm.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("fm.")})
    od.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("od.")})
    fm.to(DEV); od.to(DEV)
    og = c.OccupancyGrid(list(g["grid_after_1"].shape), float(g["step_size"]), float(g["base_threshold"]), float(g["decay"])).to(DEV)
    og.update(lambda x: od(fm(x)), jitters=cu(g["jitters"]))
    got = og.grid.cpu().numpy()
    # Which cells MAY differ from the reference's grid is computed, not typed in: a cell is decided by alpha > threshold (core.py:137-143),
    # alpha from a 3-layer MLP in fp32.  An fp64 evaluation of the same network at the golden's jittered voxel centres gives the exact
    # alpha; the distance of torch's own fp32 CPU evaluation from it (x 4) is how close to the threshold a cell must be for two fp32
    # evaluations to disagree.  Every other cell must match the golden exactly.
    from oracle import tinynerf_oracle as orc, torch_port as tp
    D, H_, W_ = g["grid_after_1"].shape
    thr, step = min(float(g["base_threshold"]), 1.0), float(g["step_size"])
    sd = {"feature_module." + k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("fm.")}
    sd.update({"sigma_decoder." + k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("od.")})

    def alpha(dtype):
        p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        out = []
        with torch.no_grad():
            for i in range(D):
                c = torch.from_numpy(orc.occupancy_voxel_coords([D, H_, W_], i, g["jitters"][i])).to(dtype)
                f = tp.mlp(p, "feature_module.net.net.", tp.posenc(c, p["feature_module.encoding.freqs"].to(dtype)))
                sgm = torch.exp(tp.mlp(p, "sigma_decoder.net.net.", f) - 1.0)
                out.append((1.0 - torch.exp(-sgm * step)).reshape(H_, W_))
        return torch.stack(out).double().numpy()
    a64, a32 = alpha(torch.float64), alpha(torch.float32)
    margin = 4.0 * float(np.abs(a32 - a64).max())
    undecided = np.abs(a64 - thr) <= margin
    flips = got != g["grid_after_1"]
    assert not (flips & ~undecided).any(), (int((flips & ~undecided).sum()), margin)
    assert undecided.mean() < 2e-3, (float(undecided.mean()), margin)          # (the fixture is not degenerate: few cells sit that close)
    assert og.mean == pytest.approx(float(g["mean_after_1"]), abs=2e-3)
    assert og.occupancy() == pytest.approx(float(g["occupancy_after_1"]), abs=2e-3)
    og.update(lambda x: od(fm(x)))                            # device-RNG path runs: cells are 1, decay or decay^2
    vals = torch.unique(og.grid)
    assert 2 <= vals.numel() <= 3 and vals.max() == 1 and 0 < (og.grid == 1).float().mean() < 1


# ------------------------------------------------------------------ RayProvider (a7)
def _provider(g, kind):
    c = core()
    og = c.OccupancyGrid(list(g["grid"].shape), 1 / 1024.).to(DEV)
    og.grid.copy_(cu(g["grid"]))
    og.mean = float(g["threshold"]) if float(g["threshold"]) < 0.01 else 1.0
    S = int(g["n_samples"])
    if kind == "aabb":
        aabb = cu(g["aabb"])
        return c.RayProvider(og, c.ContractionAABB(aabb), c.RayMarcherAABB(aabb, S, float(g["near"])))
    m = c.RayMarcherUnbounded(S, float(g["near"]), 1e5, float(g["uniform_range"]))
    t, dl = orc.unbounded_table(S, float(g["near"]), float(g["uniform_range"]))
    m._tables[str(torch.device(DEV, torch.cuda.current_device()))] = (cu(t), cu(dl))    # CPU-linspace table for bit parity
    return c.RayProvider(og, c.ContractionMip360(float("inf")), m)


@pytest.mark.parametrize("name,kind", [("G4_ray_provider_aabb", "aabb"), ("G4b_ray_provider_unbounded", "unbounded")])
def test_ray_provider_bit_exact(name, kind):
    g = load_golden(name)
    rp = _provider(g, kind)
    o, d = cu(g["rays_o"]), cu(g["rays_d"])
    packed, info = rp(o, d, training=False)
    assert info.dtype == torch.int32 and np.array_equal(info.cpu().numpy(), g["info"])
    assert np.array_equal(bits(packed), bits(g["packed"]))
    packed, info, ids = rp(o, d, training=True, jitter=cu(g["jitter"]), return_ray_ids=True)
    assert np.array_equal(info.cpu().numpy(), g["info_jit"])
    assert np.array_equal(bits(packed), bits(g["packed_jit"]))
    assert np.array_equal(ids.cpu().numpy(), np.repeat(np.arange(info.shape[0]), g["info_jit"][:, 1]))
    # device-RNG jitter: same structure, different draws
    p2, i2 = rp(o, d, training=True)
    assert i2.shape == info.shape and p2.shape[1] == 7 and int(i2[:, 1].sum()) == p2.shape[0]


def test_ray_provider_large_vs_oracle():
    """config-3 geometry at a size the oracle finishes in seconds: 800x800-style rays, 128^3 grid with
    an occupied ball, S = 256."""
    c = core()
    rng = np.random.default_rng(11)
    R, S = 2048, 256
    o = rng.standard_normal((R, 3)); o = (o / np.linalg.norm(o, axis=1, keepdims=True) * 4.0311).astype(np.float32)
    d = -o + 0.4 * rng.standard_normal((R, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, 128, dtype=np.float32)] * 3, indexing="ij")
    decay = np.float32(0.01 ** (1 / 16))
    grid = np.where(xx ** 2 + yy ** 2 + zz ** 2 < 0.25, np.float32(1), decay ** rng.integers(1, 24, (128,) * 3)).astype(np.float32)
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    jit = rng.random((R, S)).astype(np.float32)
    p_ref, i_ref = orc.ray_provider(o, d, marcher="aabb", contraction="aabb", grid=grid, threshold=0.01, n_samples=S,
                                    near=0.1, aabb=aabb, jitter=jit)
    og = c.OccupancyGrid(128, 1 / 1024.).to(DEV); og.grid.copy_(cu(grid))
    rp = c.RayProvider(og, c.ContractionAABB(cu(aabb)), c.RayMarcherAABB(cu(aabb), S, 0.1))
    p, i = rp(cu(o), cu(d), training=True, jitter=cu(jit))
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(bits(p), bits(p_ref))
    assert 0.02 < p.shape[0] / (R * S) < 0.9


def test_ray_provider_full_size_properties():
    """BASELINE config-3 size (S = 1024, B = 1024 x 8 loader batches): structure invariants."""
    c = core()
    R, S = 8192, 1024
    o = torch.nn.functional.normalize(torch.randn(R, 3, device=DEV), dim=-1) * 4.0311
    d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, device=DEV), dim=-1)
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3], device=DEV)
    og = c.OccupancyGrid(128, 1 / 1024.).to(DEV)
    og.grid[:, :, 64:] = 0.
    rp = c.RayProvider(og, c.ContractionAABB(aabb), c.RayMarcherAABB(aabb, S, 0.1))
    p, i = rp(o, d, training=True)
    cnt = i[:, 1].long()
    assert int(cnt.sum()) == p.shape[0] and (cnt <= S).all()
    assert torch.equal(i[:, 0].long(), torch.cumsum(cnt, 0) - cnt)             # exclusive scan
    assert (p[:, :3].abs() <= 1).all() and (p[:, 0] <= 1e-2).all()             # occupied half only (x<0 + interp margin)
    assert torch.equal(p[:, 3:6], torch.repeat_interleave(d, cnt, 0))          # dirs repeated per ray
    assert (p[:, 6] == float(rp.ray_marcher.step_size)).all()
    # idempotence: same seedless inference call twice -> identical packing
    p1, i1 = rp(o, d, training=False); p2, i2 = rp(o, d, training=False)
    assert torch.equal(p1, p2) and torch.equal(i1, i2)
    # empty input
    p0, i0 = rp(o[:0], d[:0], training=False)
    assert p0.shape == (0, 7) and i0.shape == (0, 2)


@pytest.mark.parametrize("scene", ["aabb", "unbounded"])
def test_coarse_reject_leaves_the_mask_unchanged(scene):
    """The block-maxima early reject (tn_occupancy_coarsen) must not change a single bit of the sampler's output:
    random grids with most values crowded around the threshold, non-multiple-of-4 sizes, rays leaving the box."""
    c = core()
    torch.manual_seed(7)
    dev = torch.device(DEV)
    for size, thr in (((37, 41, 30), 0.5), ((64, 64, 64), 0.01), ((5, 3, 9), 0.3)):
        grid = c.OccupancyGrid(size=size, step_size=0.01, threshold=thr, decay=0.9).to(dev)
        g = torch.rand(size, device=dev)
        g = torch.where(g < 0.6, thr * (1 + (torch.rand(size, device=dev) - 0.5) * 4e-3), g * 2 * thr)   # 60 % within 0.2 % of thr
        g[torch.rand(size, device=dev) < 0.3] = 0.0
        grid.grid.copy_(g)
        if scene == "aabb":
            aabb = torch.tensor([[-1., -1.2, -0.8], [1.1, 1., 0.9]], device=dev)
            prov = c.RayProvider(grid, c.ContractionAABB(aabb), c.RayMarcherAABB(aabb, 96, 0.05))
        else:
            prov = c.RayProvider(grid, c.ContractionMip360(order=float("inf")), c.RayMarcherUnbounded(96, 0.1, 1e5, uniform_range=2.0))
        o = (torch.rand(700, 3, device=dev) - 0.5) * 3
        d = torch.nn.functional.normalize(torch.randn(700, 3, device=dev), dim=-1)
        jit = torch.rand(700, 96, device=dev)
        outs = []
        for use in (True, False):
            grid.use_coarse = use
            outs.append(prov(o, d, training=True, jitter=jit))
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
        assert 0 < outs[0][0].size(0) < 700 * 96


# ------------------------------------------------------------------ harness helpers of round 3 (run.Trainer)
def test_weights_fwd_gate_flag_and_gated_mse_and_ray_gather():
    """tn_weights_fwd_gate = tn_weights_fwd + a flag raised when any weight is > 0 (never lowered); tn_mse_grad_gated = tn_mse_grad
    with grad = 0 under a closed gate; tn_gather_rays = three index_selects."""
    from tinynerf_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(5)
    info, n = ragged(rng, 300, 70, 0.1)
    it = cu(info, torch.int32)
    steps = torch.full((n,), 0.01, device=DEV)
    for scale, expect in ((5.0, 1.0), (0.0, 0.0)):                 # sigma = 0 everywhere: every weight is 0, the flag stays down
        sig = torch.rand(n, device=DEV) * scale
        w_ref = torch.empty(n, device=DEV); w = torch.empty(n, device=DEV)
        gate = torch.zeros(1, device=DEV)
        L.call("tn_weights_fwd", torch.device(DEV), L.ptr(sig), L.ptr(steps), L.ptr(it), C.c_float(1e-4), L.ptr(w_ref), C.c_int64(n), C.c_int64(info.shape[0]))
        L.call("tn_weights_fwd_gate", torch.device(DEV), L.ptr(sig), L.ptr(steps), L.ptr(it), C.c_float(1e-4), L.ptr(w), L.ptr(gate), C.c_int64(n),
               C.c_int64(info.shape[0]))
        assert torch.equal(w, w_ref) and float(gate.item()) == expect
        assert bool((w_ref > 0).any()) == bool(expect)
        # gated loss gradient
        r, t = torch.rand(1000, 3, device=DEV), torch.rand(1000, 3, device=DEV)
        g0, g1 = torch.empty_like(r), torch.empty_like(r)
        a0, a1 = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, dtype=torch.float64, device=DEV)
        L.call("tn_mse_grad", torch.device(DEV), L.ptr(r), L.ptr(t), C.c_int64(3000), C.c_float(0.25), C.c_void_p(None), L.ptr(g0), L.ptr(a0))
        L.call("tn_mse_grad_gated", torch.device(DEV), L.ptr(r), L.ptr(t), C.c_int64(3000), C.c_float(0.25), C.c_void_p(None), L.ptr(gate), L.ptr(g1), L.ptr(a1))
        assert torch.equal(g1, g0 * expect)
        assert abs(float(a0.item()) - float(a1.item())) <= 1e-12 * float(a0.item())      # (fp64 atomics: the order of the blocks is free)
    N = 5000
    o, d, c = torch.rand(N, 3, device=DEV), torch.rand(N, 3, device=DEV), torch.rand(N, 3, device=DEV)
    idx = torch.randint(0, N, (1234,), device=DEV, dtype=torch.int32)
    oo, od, oc = (torch.empty(1234, 3, device=DEV) for _ in range(3))
    L.call("tn_gather_rays", torch.device(DEV), L.ptr(o), L.ptr(d), L.ptr(c), L.ptr(idx), C.c_int64(1234), L.ptr(oo), L.ptr(od), L.ptr(oc))
    assert torch.equal(oo, o[idx.long()]) and torch.equal(od, d[idx.long()]) and torch.equal(oc, c[idx.long()])
    oc.fill_(-1.0)
    L.call("tn_gather_rays", torch.device(DEV), L.ptr(o), L.ptr(d), C.c_void_p(None), L.ptr(idx), C.c_int64(1234), L.ptr(oo), L.ptr(od), C.c_void_p(None))
    assert torch.equal(oo, o[idx.long()]) and bool((oc == -1).all())


def test_render_rays_fused_equals_two_launches():
    """tn_render_rays_fwd / _bwd (a wave per ray does weights AND composite) against the separate entry points: same weights,
    bit-identical rendered colours, same gradients (the fused backward forms d loss / d weights in registers)."""
    from tinynerf_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(11)
    dev = torch.device(DEV)
    for (n_rays, max_len, use_bg) in ((500, 90, True), (64, 1500, False), (3, 5, True)):
        info, n = ragged(rng, n_rays, max_len, 0.1)
        it = cu(info, torch.int32)
        sig = torch.rand(n, device=DEV) * 30; sig[torch.rand(n, device=DEV) < 0.2] = 0
        steps = torch.rand(n, device=DEV) * 0.02 + 0.001
        rgbs = torch.rand(n, 3, device=DEV)
        bg = torch.tensor([1.0, 0.5, 0.25], device=DEV) if use_bg else None
        R = info.shape[0]
        w0, w1 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        o0, o1 = torch.empty(R, 3, device=DEV), torch.empty(R, 3, device=DEV)
        gate = torch.zeros(1, device=DEV)
        L.call("tn_weights_fwd", dev, L.ptr(sig), L.ptr(steps), L.ptr(it), C.c_float(1e-3), L.ptr(w0), C.c_int64(n), C.c_int64(R))
        L.call("tn_composite_fwd", dev, L.ptr(rgbs), L.ptr(w0), L.ptr(it), L.ptr(bg), L.ptr(o0), C.c_void_p(None), C.c_int64(n), C.c_int64(R))
        L.call("tn_render_rays_fwd", dev, L.ptr(sig), L.ptr(steps), L.ptr(rgbs), L.ptr(it), L.ptr(bg), C.c_float(1e-3), L.ptr(w1), L.ptr(o1), L.ptr(gate),
               C.c_int64(n), C.c_int64(R))
        assert torch.equal(w0, w1) and torch.equal(o0, o1) and float(gate.item()) == 1.0
        go = torch.randn(R, 3, device=DEV)
        gr0, gw0, gs0 = torch.zeros(n, 3, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        gr1, gs1 = torch.zeros(n, 3, device=DEV), torch.zeros(n, device=DEV)
        L.call("tn_composite_bwd", dev, L.ptr(rgbs), L.ptr(w0), L.ptr(it), L.ptr(bg), L.ptr(go), L.ptr(gr0), L.ptr(gw0), C.c_int64(n), C.c_int64(R))
        L.call("tn_weights_bwd", dev, L.ptr(sig), L.ptr(steps), L.ptr(it), L.ptr(w0), L.ptr(gw0), L.ptr(gs0), C.c_int64(n), C.c_int64(R))
        L.call("tn_render_rays_bwd", dev, L.ptr(sig), L.ptr(steps), L.ptr(rgbs), L.ptr(it), L.ptr(bg), L.ptr(w0), L.ptr(go), L.ptr(gr1), L.ptr(gs1),
               C.c_int64(n), C.c_int64(R))
        assert torch.equal(gr0, gr1)
        np.testing.assert_allclose(gs1.cpu().numpy(), gs0.cpu().numpy(), rtol=0, atol=1e-6 * max(float(gs0.abs().max()), 1e-30))


@pytest.mark.parametrize("box", [1.0, 1.5, 0.3])
def test_sampler_exit_shortcut_is_exact(box):
    """Box marcher + box contraction skip the chunks behind the ray's exit (sampler.hip: exact by the monotonicity of rounding,
    only without an explicit jitter table).  training=False (no jitter, shortcut on) against training=True with a jitter table of
    zeros (t + 0 * delta = t: same candidates, shortcut off): the packed samples must be bit-identical -- boxes with power-of-two
    and other extents, rays that graze corners, run along faces, start inside the box or miss it."""
    c = core()
    rng = np.random.default_rng(int(box * 10))
    R, S = 3000, 192
    o = rng.standard_normal((R, 3)); o = (o / np.linalg.norm(o, axis=1, keepdims=True) * 2.5 * box).astype(np.float32)
    o[:200] *= 0.2                                                   # origins inside the box
    d = -o + 1.2 * box * rng.standard_normal((R, 3))
    d[200:400, 0] = 0.0                                              # axis-parallel components
    d[400:500] = (o[400:500] * [1, 0, 0]) * -1 + 1e-4 * rng.standard_normal((100, 3))   # nearly along x
    d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)).astype(np.float32)
    o[500:600, 1] = np.float32(box)                                  # origins on a face plane, grazing rays
    aabb = cu(np.array([[-box] * 3, [box] * 3], np.float32))
    og = c.OccupancyGrid(64, 1 / 1024.).to(DEV)
    og.grid.copy_(torch.rand(64, 64, 64, device=DEV) * 0.05)         # ~80 % of the cells above the 0.01 threshold
    rp = c.RayProvider(og, c.ContractionAABB(aabb), c.RayMarcherAABB(aabb, S, 0.05))
    p0, i0 = rp(cu(o), cu(d), training=False)
    p1, i1 = rp(cu(o), cu(d), training=True, jitter=torch.zeros(R, S, device=DEV))
    assert torch.equal(i0, i1) and np.array_equal(bits(p0), bits(p1))
    assert 0.05 < p0.shape[0] / (R * S) < 0.95


@pytest.mark.parametrize("R,S,box", [(2048, 256, 1.5), (1024, 1024, 1.5), (700, 192, 1.0)])
def test_production_sampler_path_device_rng_vs_oracle_bits(R, S, box):
    """Round-3 verdict: the path training actually takes -- jitter from the device's counter RNG, the exit shortcut on (no explicit
    table), the coarse early reject on -- had only HIP-vs-HIP coverage.  The RNG is restated in the oracle (``orc.uniform01``,
    pinned by known answers on the CPU); the oracle's ``ray_provider`` with THAT jitter table (reference core.py:165-188,
    training=True) must give the same ``packing_info`` and the same packed coordinates / directions / steps, bit for bit:
    2048 x 256, S = 1024 (the BASELINE shape) and a box with power-of-two extents; cameras outside the box looking in, as in training."""
    from oracle import tinynerf_oracle as orc
    c = core()
    rng = np.random.default_rng(R + S)
    o = rng.standard_normal((R, 3)); o = (o / np.linalg.norm(o, axis=1, keepdims=True) * 2.7 * box).astype(np.float32)
    d = -o + 0.8 * box * rng.standard_normal((R, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[:50] *= 0.1                                                                  # some origins inside the box
    d[50:80, 1] = 0.0                                                              # a zero direction component (core.py:75)
    aabb_np = np.array([[-box] * 3, [box] * 3], np.float32)
    lin = np.linspace(-1, 1, 64, dtype=np.float32)
    zz, yy, xx = np.meshgrid(lin, lin, lin, indexing="ij")
    grid_np = np.where(xx * xx + yy * yy + zz * zz < 0.36, 1.0, 0.74989 ** 20).astype(np.float32)     # the bench's ball, decayed outside
    grid_np *= (0.6 + 0.4 * rng.random(grid_np.shape, dtype=np.float32))                               # values on both sides of the threshold
    og = c.OccupancyGrid(64, 1 / 1024.).to(DEV)
    og.grid.copy_(cu(grid_np))
    og.mean = float(og.grid.mean().item())
    assert og.use_coarse                                                           # production setting
    rp = c.RayProvider(og, c.ContractionAABB(cu(aabb_np)), c.RayMarcherAABB(cu(aabb_np), S, 0.1))
    torch.manual_seed(4242 + S)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())                             # what RayProvider._desc will draw
    torch.manual_seed(4242 + S)
    packed, info = rp(cu(o), cu(d), training=True)
    jit = orc.sampler_jitter(seed, R, S)
    ref_packed, ref_info = orc.ray_provider(o, d, marcher="aabb", contraction="aabb", grid=grid_np, threshold=float(og.threshold), n_samples=S,
                                            near=0.1, aabb=aabb_np, jitter=jit)
    assert np.array_equal(info.cpu().numpy(), ref_info)
    assert np.array_equal(bits(packed), ref_packed.view(np.int32))
    frac = ref_packed.shape[0] / (R * S)
    assert 0.01 < frac < 0.6, frac
    # ... and the jitter did something: the unjittered pass keeps a different set
    p0, i0 = rp(cu(o), cu(d), training=False)
    assert not torch.equal(i0, info)
