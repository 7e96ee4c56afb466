"""N > 1 on the GPU (the reference is single-device, run.py:98, so the oracle here is the 1-rank path):

* the split backward the N > 1 step uses -- data gradients (TN_MLP_CHAIN_ONLY) -> plane scatter -> weight gradients
  (TN_MLP_WGRAD_ONLY) -- against the one-shot tn_mlp_bwd_pair;
* two ranks sharing one GPU (gloo rendezvous on 127.0.0.1, CUDA tensors) run the real ``Trainer.step()`` on disjoint
  halves of a ray set; loss, every reduced ``param.grad`` before Adam and the occupancy grids equal one rank that
  processes the union of the two ranks' batches;
* the same exchange path over the real backend: a one-rank RCCL group on this GPU under a Trainer told world_size = 2 (every
  collective issued and awaited as on 8 GPUs, every sum the identity) equals the plain 1-rank step.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_split_backward_equals_one_shot():
    from tinynerf_amd import _lib as L
    from tinynerf_amd import models as m
    from tinynerf_amd.models import _hwc, _kplanes_desc, _mlp_desc
    torch.manual_seed(33)
    n, R, F = 6007, 83, 96
    dev = torch.device(DEV)
    sp = [p.detach().contiguous() for p in m.VanillaOpacityDecoder(F).to(dev).net.params()]
    cd = m.VanillaColorDecoder(8, F, 64, 3).to(dev)
    rp = [p.detach().contiguous() for p in cd.net.params()]
    field = m.KPlanesFeatureField(32, (16, 32, 64)).to(dev)
    planes = field.plane_tensors()
    coords = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    kdesc, keep = _kplanes_desc(planes)
    x = torch.empty(n, F, device=dev)
    L.call("tn_kplanes_fwd", dev, C.byref(kdesc), L.ptr(coords), C.c_int64(3), C.c_int64(n), L.ptr(x))
    ray_ids = torch.sort(torch.randint(0, R, (n,), device=dev, dtype=torch.int32)).values.contiguous()
    dirs_ray = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    table = torch.empty(R, 56, device=dev)
    L.call("tn_dir_encode", dev, L.ptr(dirs_ray), C.c_int64(R), L.ptr(cd.pe.freqs), C.c_int(8), L.ptr(table), C.c_int(56))
    g_rgb, g_sig = torch.randn(n, 3, device=dev), torch.randn(n, 1, device=dev)
    fn = L.lib().tn_mlp_bwd_workspace_bytes
    fn.restype = C.c_int64
    rd = _mlp_desc(rp, F, L.ENC_AUX_CAT, 8, L.ACT_SIGMOID, cd.pe.freqs, 0, ray_ids, 56)
    sd = _mlp_desc(sp, F, L.ENC_NONE, 0, L.ACT_EXP_M1, None, 0, None, 0)
    nbr, nbs = int(fn(C.byref(rd), C.c_int64(n))), int(fn(C.byref(sd), C.c_int64(n)))
    wr0, ws0 = torch.empty(nbr // 4, device=dev), torch.empty(nbs // 4, device=dev)
    rgb, sigma = torch.empty(n, 3, device=dev), torch.empty(n, 1, device=dev)
    L.call("tn_mlp_fwd_stash_pair", dev, C.byref(rd), C.byref(sd), L.ptr(x), L.ptr(table), C.c_int64(n), L.ptr(rgb), L.ptr(sigma),
           L.ptr(wr0), C.c_int64(nbr), L.ptr(ws0), C.c_int64(nbs))
    sd.flags = L.MLP_STASHED

    def run(split: bool):
        wr, ws = wr0.clone(), ws0.clone()
        grs, gss = [torch.zeros_like(p) for p in rp], [torch.zeros_like(p) for p in sp]
        gpl = [torch.zeros_like(p) for p in planes]
        arr = lambda gs, o: (C.c_void_p * (len(gs) // 2))(*[g.data_ptr() for g in gs[o::2]])
        gp = ((C.c_void_p * 3) * L.TN_KPLANES_MAX_SCALES)()
        for s in range(3):
            for p in range(3):
                gp[s][p] = _hwc(gpl[3 * s + p]).data_ptr()
        gx = torch.full((n, F), float("nan"), device=dev)
        args = (C.byref(sd), L.ptr(x), L.ptr(table), L.ptr(g_rgb), L.ptr(g_sig), C.c_int64(n), arr(grs, 0), arr(grs, 1), arr(gss, 0),
                arr(gss, 1), L.ptr(gx), L.ptr(wr), C.c_int64(nbr), L.ptr(ws), C.c_int64(nbs))
        if split:
            rd.flags = L.MLP_STASHED | L.MLP_CHAIN_ONLY
            L.call("tn_mlp_bwd_pair", dev, C.byref(rd), *args)
            assert all(float(g.abs().max()) == 0.0 for g in grs + gss)       # no weight gradient yet
            L.call("tn_kplanes_bwd", dev, C.byref(kdesc), L.ptr(coords), C.c_int64(3), C.c_int64(n), L.ptr(gx), gp)
            rd.flags = L.MLP_STASHED | L.MLP_WGRAD_ONLY
            L.call("tn_mlp_bwd_pair", dev, C.byref(rd), *args)
        else:
            rd.flags = L.MLP_STASHED
            L.call("tn_mlp_bwd_pair", dev, C.byref(rd), *args)
            L.call("tn_kplanes_bwd", dev, C.byref(kdesc), L.ptr(coords), C.c_int64(3), C.c_int64(n), L.ptr(gx), gp)
        return gx, grs + gss, gpl

    gx0, g0, p0 = run(False)
    gx1, g1, p1 = run(True)
    assert torch.equal(gx0, gx1)                                             # same kernel, same arithmetic: bit-equal
    for a_, b_ in zip(g1, g0):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, float(b_.abs().max())))
    for a_, b_ in zip(p1, p0):                                               # atomics: order differs, values agree
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=0, atol=1e-5 * max(1.0, float(b_.abs().max())))


# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene():
    from tinynerf_amd import rays
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=4, res=64, seed=7, device="cpu")
    return o.contiguous(), d.contiguous(), rgb.contiguous()


def _cfg(method, sharded=False):
    from tinynerf_amd.run import TrainConfig
    return TrainConfig(method=method, scene_type="aabb", batch_size=256, n_samples=48, seed=4, occupancy_res=32, deterministic=True,
                       kplanes_resolutions=(32, 64, 128),        # 128^2 x 32 >= 2^18 elements: its own early all-reduce
                       sharded_optimizer=sharded)                # True: reduce-scatter -> Adam on the rank's rows -> all-gather


N_STEPS = 3


def _half_empty_grid(tr):
    """an occupancy grid with empty borders and no refresh during the test: the plane-gradient exchange of the 2-rank run is then
    restricted to the live rows (Trainer._refresh_reduce_rows), and must still equal the 1-rank result everywhere"""
    g = tr.occupancy_grid
    g.grid.zero_()
    g.grid[6:22, 9:27, 4:30] = 1.0
    g.mean = float(g.grid.mean().item())
    tr.occupancy_grid_updates = 10 ** 9
    tr.train_step = 1
    tr._refresh_reduce_rows(); tr._refresh_reduce_rows()
    if tr.world > 1 and tr._plane_of:
        assert all(r1 - r0 < p.size(2) for (r0, r1), p in zip(tr._reduce_rows, tr.renderer.feature_module.plane_tensors()))


def _ulp_perturbed(tr, seed):
    """a control run: every parameter moved by a few ulps (x (1 + 2^-22 (U - 0.5))).  Two code paths that differ by one rounding anywhere -- the
    MSE scale as a device scalar instead of a host float, a sum taken in another order -- are this far apart after Adam (eps 1e-15: the update of a
    near-zero gradient element is +- lr whatever its size) has fed the difference back; run-to-run repeats of ONE path are not (they are nearly
    bit-equal and say nothing about that amplification)."""
    gen = torch.Generator(device=tr.device).manual_seed(seed)
    with torch.no_grad():
        for p_ in tr.renderer.parameters():
            p_.mul_(1.0 + 2.0 ** -22 * (torch.rand(p_.shape, device=p_.device, generator=gen) - 0.5))


def _no_dropout(tr):
    """Cobafa's Dropout(0.01) (models.py:250) draws from the process RNG: switched off so that 2 ranks and 1 rank see the same function"""
    drop = getattr(tr.renderer.feature_module, "dropout", None)
    if drop is not None:
        drop.p = 0.0


def _grads(t):
    """every reduced gradient on step 0; later steps skip the multi-million-element grids (queue traffic), keep all the rest"""
    return {k: p.grad.detach().cpu().contiguous().numpy().copy() for k, p in t.renderer.named_parameters()
            if t.train_step == 1 or p.numel() < (1 << 20)}          # numpy: pickled through the queue, no shared-memory handles


def _smooth_adam(tr, smooth):
    """Adam with eps 1e-15 (run.py:186) turns the order noise of a near-zero gradient element (1e-5 of the largest one) into a
    full-size update -- the reason the recipe's later steps can only be compared loosely.  `smooth`: the SAME update rule on both
    sides with eps = 1 (against gradients scaled by 2^10: close to plain SGD, Lipschitz in the gradient), so that steps 1 and 2
    compare the exchange as tightly as step 0 does."""
    if smooth:
        for g in tr.optimizer.param_groups:
            g["eps"] = 1.0


def _rank_main(rank, world, port, method, q, smooth=False, sharded=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tinynerf_amd.run import Trainer
    dev = torch.device(DEV, 0)
    torch.cuda.set_device(dev)
    o, d, rgb = _scene()
    half = slice(rank, None, world)                                 # disjoint halves of the ray set
    tr = Trainer(_cfg(method, sharded), o[half].to(dev), d[half].to(dev), rgb[half].to(dev), torch.ones(3, device=dev), dev, rank=rank,
                 world_size=world)
    assert tr._sharded == (sharded and method == "kplanes")
    _no_dropout(tr)
    _half_empty_grid(tr)
    _smooth_adam(tr, smooth)
    cap = {}
    tr.grad_hook = lambda t: cap.__setitem__("g", _grads(t))
    early_calls = [0]
    planes_ready = tr._planes_ready

    def counted(grads):
        early_calls[0] += 1
        planes_ready(grads)
    tr._planes_ready = counted
    out = []
    for _ in range(N_STEPS):
        cursor = tr._cursor
        st = tr.step()
        out.append(dict(cursor=cursor, n_rays=int(st["n_rays"]), n_samples=int(st["n_samples"]), loss=tr.loss_value(), grads=cap["g"],
                        grid=tr.occupancy_grid.grid.cpu().numpy().copy(), pending=len(tr._early), early_calls=early_calls[0],
                        params={k: p.detach().cpu().contiguous().numpy().copy() for k, p in tr.renderer.named_parameters()
                                if sharded and (tr.train_step == 1 or p.numel() < (1 << 20))}))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("method,world,smooth,sharded", [("kplanes", 2, False, False), ("cobafa", 2, False, False), ("kplanes", 4, True, False),
                                                         ("cobafa", 2, True, False), ("kplanes", 2, True, True), ("kplanes", 4, True, True)])
def test_ranks_equal_one_rank_on_the_union(method, world, smooth, sharded):
    """`world` gloo ranks sharing this GPU run the real Trainer.step() on disjoint shares of a ray set; one rank on the union of
    their batches must give the same loss, reduced gradients and occupancy grids.  smooth=False: the reference's optimizer
    (later steps loose, see _smooth_adam); smooth=True: every step as tight as the first.  world = 4: the exchange code with
    more than one peer (bucket + gate slot, coalesced live-row slices, rank-strided ray streams).  sharded (round 5): the plane
    gradients travel as reduce-scatter, every rank runs Adam + TV on ITS rows of every plane and the updated rows are all-gathered --
    a rank's reduced plane gradient is then only defined on its own rows, and what must agree with the one-rank run (and between the
    ranks, bit for bit) are the PARAMETERS after the step."""
    from tinynerf_amd.run import Trainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, method, q, smooth, sharded)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)

    dev = torch.device(DEV, 0)
    o, d, rgb = _scene()

    def one_rank(perturb=0):
        t = Trainer(_cfg(method), o.to(dev), d.to(dev), rgb.to(dev), torch.ones(3, device=dev), dev)
        _no_dropout(t)
        _half_empty_grid(t)
        _smooth_adam(t, smooth)
        if perturb:
            _ulp_perturbed(t, perturb)
        c = {}
        t.grad_hook = lambda t_, c=c: c.__setitem__("g", _grads(t_))
        return t, c
    tr, cap = one_rank()
    # Tolerance by construction: two more one-rank runs on the same batches from parameters moved by a few ulps ("controls",
    # _ulp_perturbed).  How far THEY are from the first one -- one rounding's worth of difference, amplified by Adam from the second step
    # on -- is the noise of this tensor at this step; the ranks may be 4 x the larger of the two distances away, or the typed-in bound
    # below where the controls agree better than that.
    controls = [one_rank(201), one_rank(202)]
    for step in range(N_STEPS):
        os_, ds_, ts_ = [], [], []
        for rank in range(world):                                   # the rays each rank consumed in this step
            r = res[rank][step]
            oh, dh, th = o[rank::world], d[rank::world], rgb[rank::world]
            idx = (r["cursor"] + torch.arange(r["n_rays"])) % oh.size(0)
            os_.append(oh[idx]); ds_.append(dh[idx]); ts_.append(th[idx])
        ou, du, tu = torch.cat(os_).to(dev), torch.cat(ds_).to(dev), torch.cat(ts_).to(dev)
        packed, info = tr.ray_provider(ou, du, training=False)
        assert packed.size(0) == sum(res[rank][step]["n_samples"] for rank in range(world))
        tr.step_on_batch(packed, info, tu, prefetch=False)
        loss = tr.loss_value()
        loss_noise, noise_norm, noise_max = 0.0, {}, {}
        for tc, cc in controls:
            pc, ic = tc.ray_provider(ou, du, training=False)
            tc.step_on_batch(pc, ic, tu, prefetch=False)
            loss_noise = max(loss_noise, abs(tc.loss_value() - loss))
            for k, ref in cap["g"].items():
                dlt = (cc["g"][k] - ref).astype(np.float64)
                noise_norm[k] = max(noise_norm.get(k, 0.0), float(np.linalg.norm(dlt)))
                noise_max[k] = max(noise_max.get(k, 0.0), float(np.abs(dlt).max()))
        # first step: the same per-sample arithmetic on both sides, summed by atomics in another order (a grid voxel of the
        # half-empty scene collects thousands of terms: 1e-4 of the largest element); later steps: Adam (eps 1e-15) amplifies it
        first = step == 0 or smooth             # tolerance class of the step (see _smooth_adam)
        tol = 1e-4 if first else 2e-3
        for rank in range(world):
            r = res[rank][step]
            assert abs(r["loss"] - loss) <= max(tol * abs(loss), 4.0 * loss_noise), (step, rank, r["loss"], loss, loss_noise)
            for k, ref in cap["g"].items():
                got = r["grads"][k]
                nn_, nm_ = noise_norm[k], noise_max[k]      # (of the whole tensor: an upper bound for a row slice's norm, the slice's maximum at most)
                if sharded and ".plane" in k:           # reduce-scatter: the rank holds the sum on ITS rows of the plane only
                    r0, r1 = Trainer._own_rows(ref.shape[2], rank, world)
                    got, ref = got[:, :, r0:r1], ref[:, :, r0:r1]
                if ref.size >= (1 << 16) or ".plane" in k:           # (a rank's row slice of a small plane is still a sum of atomics)
                    # a grid voxel / plane texel sums thousands of atomics whose terms cancel (+-1e-3 summing to 1e-4): single
                    # elements carry 1e-4 of the largest element as order noise, and from the second step on Adam (eps 1e-15)
                    # turns that noise into full-size updates of the elements it hits; the tensor as a whole must agree to
                    # 2e-5 on the first step and to 5e-3 afterwards
                    # (smooth steps after the first: the parameters themselves differ by the first step's order noise -- seen up to 2.8e-5)
                    norm_tol = 2e-5 if step == 0 else (1e-4 if smooth else 5e-3)
                    assert float(np.linalg.norm((got - ref).astype(np.float64))) <= max(norm_tol * float(np.linalg.norm(ref.astype(np.float64))), 4.0 * nn_), (k, step, nn_)
                    np.testing.assert_allclose(got, ref, rtol=0, atol=max((5e-3 if first else 5e-2) * max(float(np.abs(ref).max()), 1e-12), 4.0 * nm_), err_msg=k)
                else:
                    # MLP weight gradients.  Typical distance: 1e-6 of the largest element on every step (two ranks sum two halves of the
                    # samples where one rank sums all of them).  The rare outlier is discrete: ONE sample's hidden unit within rounding of
                    # zero takes the other ReLU branch ("ReLU ties") and the elements of that unit's row move by that sample's
                    # share of a sum over ~12 k samples -- seen: 4 of 4096 elements of one head at 1.25e-4 on one run in eleven; the ulp
                    # controls flip ties as often, but rarely in the same run.  So: the tensor as a whole to `tol`, single elements to 10 x.
                    assert float(np.linalg.norm((got - ref).astype(np.float64))) <= max(tol * float(np.linalg.norm(ref.astype(np.float64))), 4.0 * nn_), (k, step, nn_)
                    np.testing.assert_allclose(got, ref, rtol=0, atol=max(10.0 * tol * max(float(np.abs(ref).max()), 1e-12), 4.0 * nm_), err_msg=k)
            if step == 0:
                assert np.array_equal(r["grid"], tr.occupancy_grid.grid.cpu().numpy())      # identical grids without communication
            if sharded:
                # parameters after the optimizer pass + all-gather: identical on every rank, and the one-rank run's up to the order noise of
                # the gradients (Adam with eps = 1: Lipschitz) -- every row of every plane, i.e. also the rows other ranks updated
                for k, pv in r["params"].items():
                    assert np.array_equal(pv, res[0][step]["params"][k]), (k, step, rank)
                    one = dict(tr.renderer.named_parameters())[k].detach().cpu().contiguous().numpy()
                    pn = max(float(np.linalg.norm((dict(tc.renderer.named_parameters())[k].detach().cpu().contiguous().numpy() - one).astype(np.float64)))
                             for tc, _ in controls)
                    assert float(np.linalg.norm((pv - one).astype(np.float64))) <= max(2e-5 * float(np.linalg.norm(one.astype(np.float64))), 4.0 * pn), (k, step, pn)
        assert all(np.array_equal(res[0][step]["grid"], res[rank][step]["grid"]) for rank in range(1, world))
    assert not any(r["pending"] for rank in range(world) for r in res[rank])       # every early all-reduce was awaited
    if method == "kplanes":      # the fused node handed its plane gradients over mid-backward (CHAIN_ONLY -> scatter -> WGRAD_ONLY) every step
        assert all(res[rank][-1]["early_calls"] == N_STEPS for rank in range(world))


# ------------------------------------------------------------------------------------------------------------------
def _rccl_main(port, method, q):
    """The N > 1 step over the REAL backend: a one-rank RCCL ("nccl") group on this GPU, a Trainer told world_size = 2 -- every
    collective of the exchange path (global ray count + gate, early per-plane all-reduces started inside the backward pass on
    RCCL's stream while the weight-gradient kernels run, the bucket, the loss) is issued and awaited exactly as on 8 GPUs; with
    one rank in the group each sum is the identity, so the result must equal the plain 1-rank step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", TORCH_NCCL_HIGH_PRIORITY="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device(DEV, 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from tinynerf_amd.run import Trainer
    o, d, rgb = _scene()
    out = {}
    # "control*": the plain step three more times from parameters moved by a few ulps (_ulp_perturbed)
    for key, world in (("control0", 1), ("control1", 1), ("control2", 1), (1, 1), (2, 2)):
        tr = Trainer(_cfg(method), o.to(dev), d.to(dev), rgb.to(dev), torch.ones(3, device=dev), dev, rank=0, world_size=world)
        _no_dropout(tr)
        _half_empty_grid(tr)
        if isinstance(key, str):
            _ulp_perturbed(tr, 100 + int(key[-1]))
        cap = {}
        tr.grad_hook = lambda t, cap=cap: cap.__setitem__("g", {k: p.grad.detach().clone() for k, p in t.renderer.named_parameters()})
        calls = [0]
        if world > 1:
            ready = tr._planes_ready

            def counted(grads, ready=ready, calls=calls):
                calls[0] += 1
                ready(grads)
            tr._planes_ready = counted
        steps = []
        for _ in range(N_STEPS):
            st = tr.step()
            steps.append((int(st["n_samples"]), {k: v.cpu().numpy() for k, v in cap["g"].items()}, len(tr._early)))
        torch.cuda.synchronize()
        out[key] = dict(steps=steps, params={k: p.detach().cpu().numpy() for k, p in tr.renderer.named_parameters()}, early_calls=calls[0])
    q.put(out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("method", ["kplanes", "cobafa"])
def test_exchange_path_over_rccl_with_one_rank(method):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_main, args=(_free_port(), method, q))
    p.start()
    out = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    one, two, ctls = out[1], out[2], [out["control0"], out["control1"], out["control2"]]
    for step in range(N_STEPS):
        assert one["steps"][step][0] == two["steps"][step][0]                     # same batches
        assert two["steps"][step][2] == 0                                         # every early all-reduce was awaited
        for k, ref in one["steps"][step][1].items():
            got = two["steps"][step][1][k]
            # Same kernels on the same batch; the MSE scale is a device scalar instead of a host float, plane / grid sums are atomics in
            # whatever order the waves arrive.  Tolerance by construction: three control runs of the plain step from parameters moved by a
            # few ulps say how far two evaluations that differ by one rounding are apart for THIS tensor at THIS step -- a grid whose
            # gradient is a near-cancelling sum (norm 2e-5) moves by percents of its own norm, and from step 1 on Adam (eps 1e-15) feeds
            # that back -- and the exchange path may be 4 x the largest of the three distances from the plain one; where the controls
            # agree better, 2e-5 of the tensor's norm is asked.  (Plain repeats of one path are nearly bit-equal and bound nothing: with
            # them as controls the step-1 gradient of a Cobafa basis grid, 1.07e-4 of its norm away, failed on half of the boxes.)
            nrm = float(np.linalg.norm(ref.astype(np.float64)))
            noise = max(float(np.linalg.norm((c["steps"][step][1][k] - ref).astype(np.float64))) for c in ctls)
            assert float(np.linalg.norm((got - ref).astype(np.float64))) <= max(2e-5 * nrm, 4.0 * noise) + 1e-30, (k, step, nrm, noise)
    for k, ref in one["params"].items():
        noise = max(float(np.abs(c["params"][k] - ref).max()) for c in ctls)
        np.testing.assert_allclose(two["params"][k], ref, rtol=0, atol=max(2e-5 * float(np.abs(ref).max()), 4.0 * noise) + 1e-12, err_msg=k)
    if method == "kplanes":
        assert two["early_calls"] == N_STEPS


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(600)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launcher_relays_two_ranks(scaling):
    """`python bench.py --gpus 2` without a launcher: the parent (which must not touch the GPU: it never imports torch) starts two
    ranks; here they share this GPU over gloo (TN_BENCH_BACKEND, debug).  The relayed JSON line must describe the 2-rank job."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TN_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "1",
                        "--views", "2", "--no-cpu-baseline", "--no-stages", "--scaling", scaling], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == scaling
    assert line["config"]["parallelism"].startswith("dp2") and "gloo" in line["config"]["parallelism"] and scaling in line["config"]["parallelism"]
    per_gpu = line["config"]["samples_per_step_per_gpu"]
    # weak: every rank runs the recipe's batch (~2^20 samples per rank and step); strong: the recipe's 2^20 are split over the ranks
    assert line["value"] > 0 and (0.9 * 2 ** 20 < per_gpu < 1.3 * 2 ** 20 if scaling == "weak" else 0.9 * 2 ** 19 < per_gpu < 1.3 * 2 ** 19), per_gpu
    assert line["psnr_at_step"]["step"] == 3 and 5.0 < line["psnr_at_step"]["psnr"] < 40.0
    assert line["refresh"]["every_steps"] == 64 and line["refresh"]["ms_per_refresh"] > 0 and line["value_with_refresh"] < line["value"]
    assert "cpu_baseline" not in line                     # rank 0 at N = 1 only
