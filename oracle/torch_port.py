"""Differentiable CPU port of the render path, op for op what the reference executes.

TEST INFRASTRUCTURE ONLY (see oracle/tinynerf_oracle.py): used by tests/ as the gradient oracle at
sizes where golden vectors would be too large, and by ``bench.py``'s ``cpu_baseline`` leg, where it
stands for "the reference's CPU path" (the reference itself has no CPU weights kernel).

Functional restatement with torch CPU ops -- the same ATen primitives the reference calls
(grid_sample / linear / index_add_; reference src/models.py:93-163,7-89 and src/core.py:225-267) --
plus the C restatement of the weights kernels (oracle/weights_ref.c) wrapped in an autograd Function.
Parameters are passed as a flat dict with the reference's state_dict keys.
"""
from __future__ import annotations

import itertools
from typing import Dict, Optional

import numpy as np
import torch

from . import tinynerf_oracle as orc

PAIRS = list(itertools.combinations(range(3), 2))        # models.py:146


def plane_lookup(plane: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """models.py:105-113: bilinear, zeros padding, align_corners=True -> [N,C]."""
    out = torch.nn.functional.grid_sample(plane, xy.view(1, -1, 1, 2), align_corners=True)
    return out.view(plane.size(1), -1).t()


def kplanes_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "feature_module.") -> torch.Tensor:
    """models.py:153-163."""
    feats = []
    s = 0
    while f"{prefix}planes.{s}.0.plane" in sd:
        prod = None
        for p, (i, j) in enumerate(PAIRS):
            v = plane_lookup(sd[f"{prefix}planes.{s}.{p}.plane"], x[:, (i, j)])
            prod = v if prod is None else prod * v
        feats.append(prod)
        s += 1
    return torch.cat(feats, -1)


def _grid_lookup(grid: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """models.py:226-232 (CobafaGrid.forward): trilinear, zeros padding, align_corners=True -> [N,C]."""
    out = torch.nn.functional.grid_sample(grid, pts.reshape(1, -1, 1, 1, 3), align_corners=True)
    return out.view(grid.size(1), -1).t()


def cobafa_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, freqs, prefix: str = "feature_module.", dropout_p: float = 0.0) -> torch.Tensor:
    """models.py:252-266: coefficient grid x sawtooth-warped basis grids -> concat -> (dropout) -> MLP.  ``freqs`` are the
    SawtoothEncoding constants (plain Python attributes in the reference, not in the state_dict)."""
    coefs = _grid_lookup(sd[prefix + "coef_grid.grid"], x)
    feats = []
    for i, f in enumerate(freqs):
        warped = 2. * ((f * x) % 1.) - 1.                                      # models.py:213-215
        feats.append(_grid_lookup(sd[prefix + f"basis_grids.{i}.grid"], warped) * coefs[:, [i]])
    feat = torch.cat(feats, -1)
    if dropout_p > 0:
        feat = torch.nn.functional.dropout(feat, dropout_p, True)
    return mlp(sd, prefix + "mlp.net.", feat)


def _layers(sd: Dict[str, torch.Tensor], prefix: str):
    keys = sorted({k[len(prefix):].rsplit(".", 1)[0] for k in sd if k.startswith(prefix) and k.endswith(".weight")},
                  key=lambda s: int(s.split(".")[0]))
    return [(sd[prefix + k + ".weight"], sd[prefix + k + ".bias"]) for k in keys]


class ReluControl:
    """Bookkeeping for hidden units whose pre-activation is an fp32 tie: |pre| <= eps * (|x| . |w| + |b|), i.e. the sign of
    the sum depends on the summation order (ATen's blocked sgemm vs the MFMA K-order of the HIP kernels).  Such a unit's ReLU
    may legitimately be on in one implementation and off in the other; its whole backward contribution then differs although
    both are correct fp32 evaluations.  While an instance is installed (``with ctrl:``) every ``mlp`` call records the ties it
    meets in ``found`` as (prefix, layer, row, column) (and in ``state`` the ReLU state this evaluation took plus |pre| / magnitude) and applies the states listed in ``force`` {(prefix, layer, row, column):
    bool}.  Used by tests/_ties.py to compare gradients "up to the state of the tie units" for ANY seed."""

    current: Optional["ReluControl"] = None

    def __init__(self, eps: float = 4e-6, force: Optional[dict] = None):
        self.eps, self.force, self.found, self.state = eps, dict(force or {}), [], {}

    def __enter__(self):
        self._prev, ReluControl.current = ReluControl.current, self
        return self

    def __exit__(self, *exc):
        ReluControl.current = self._prev

    def relu(self, prefix: str, li: int, pre: torch.Tensor, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            mag = torch.nn.functional.linear(x.abs(), w.abs(), b.abs())
            near = torch.nonzero(pre.abs() <= self.eps * mag)
            mask = pre > 0
            for r, c in near.tolist():
                self.found.append((prefix, li, r, c))
                self.state[(prefix, li, r, c)] = (bool(mask[r, c]), float(pre[r, c].abs() / mag[r, c]))
                if (prefix, li, r, c) in self.force:
                    mask[r, c] = self.force[(prefix, li, r, c)]
        return pre * mask


def mlp(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """models.py:7-28."""
    layers = _layers(sd, prefix)
    ctrl = ReluControl.current
    for li, (w, b) in enumerate(layers):
        pre = torch.nn.functional.linear(x, w, b)
        if li + 1 == len(layers):
            return pre
        x = torch.relu(pre) if ctrl is None else ctrl.relu(prefix, li, pre, x, w, b)
    return x


def posenc(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """models.py:36-39."""
    a = x[..., None] * freqs
    return torch.cat([torch.sin(a), torch.cos(a)], -1).flatten(-2)


class _TruncExp(torch.autograd.Function):
    """models.py:42-53."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(torch.clamp(ctx.saved_tensors[0], min=-15, max=15))


def explicit_sigma(sd: Dict[str, torch.Tensor], feat: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """models.py:183-191 (KPlanesExplicitOpacityDecoder): exp(<f, W f + b> - 1) with the truncated exponential."""
    basis = torch.nn.functional.linear(feat, sd[prefix + "net.weight"], sd[prefix + "net.bias"])
    return _TruncExp.apply(torch.sum(feat * basis, -1, keepdim=True) - 1.)


def explicit_rgb(sd: Dict[str, torch.Tensor], feat: torch.Tensor, dirs: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """models.py:193-205 (KPlanesExplicitColorDecoder): sigmoid(<f, B_k>) with B = MLP([PE(d), d, f]).view(n, 3, C)."""
    inp = torch.cat([posenc(dirs, sd[prefix + "pe.freqs"]), dirs, feat], -1)
    basis = mlp(sd, prefix + "net.net.", inp).view(-1, 3, feat.size(-1))
    return torch.sigmoid(torch.sum(feat.unsqueeze(-2) * basis, -1))


class _Weights(torch.autograd.Function):
    """core.py:192-207 over the C restatement of cuda.cu.  ``exact_backward`` (tests only): the same formula in fp64 without the
    cancellation (orc.weights_bwd_fp64) -- the distance between the two results is the conditioning of a fixture."""
    exact_backward = False
    noise_probe = None              # tests only: seed of a rounding-noise draw added to the backward's result (weights_conditioning)

    @staticmethod
    def forward(ctx, sigmas, steps, info, thr):
        w = torch.from_numpy(orc.weights_fwd(sigmas.detach().numpy(), steps.detach().numpy(), info.numpy(), float(thr)))
        ctx.save_for_backward(sigmas, steps, info, w)
        return w

    @staticmethod
    def backward(ctx, g):
        s, d, info, w = ctx.saved_tensors
        fn = orc.weights_bwd_fp64 if _Weights.exact_backward else orc.weights_bwd
        gs = fn(s.detach().numpy(), d.detach().numpy(), info.numpy(), w.numpy(), g.contiguous().numpy())
        if _Weights.noise_probe is not None:
            # cuda.cu:49-56 forms every suffix sum as (-total + prefix) in fp32, two sequential passes of `count` additions:
            # each suffix carries an absolute error of about 2^-23 * sqrt(count) * sum_j |w_j g_j| of its ray (measured: the
            # same reference code on two hosts -- different expf -- differs by 2.4e-4 on G9's sigma head, the wave-scan kernel
            # by 4e-4), and the result multiplies it by the step size
            rng = np.random.default_rng(_Weights.noise_probe)
            wg = np.abs(w.numpy() * g.contiguous().numpy())
            scale = np.zeros_like(gs)
            for a, c in info.numpy():
                scale[a:a + c] = wg[a:a + c].sum() * np.sqrt(max(int(c), 1))
            gs = (gs + d.detach().numpy() * scale * np.float32(2.0 ** -23) * rng.uniform(-1, 1, gs.shape).astype(np.float32)).astype(np.float32)
        return torch.from_numpy(gs), None, None, None


def features(sd: Dict[str, torch.Tensor], x: torch.Tensor, vanilla_freqs: int = 0, cobafa_freqs=None) -> torch.Tensor:
    """the three feature modules of run.py:130-147 behind one call"""
    if cobafa_freqs is not None:
        return cobafa_features(sd, x, cobafa_freqs)
    if vanilla_freqs:
        return mlp(sd, "feature_module.net.net.", posenc(x, sd["feature_module.encoding.freqs"]))
    return kplanes_features(sd, x)


def render(sd: Dict[str, torch.Tensor], packed: torch.Tensor, info: torch.Tensor, bg: Optional[torch.Tensor],
           thr: float = 1e-4, vanilla_freqs: int = 0, cobafa_freqs=None) -> torch.Tensor:
    """core.py:225-267 for a K-Planes (or Vanilla: vanilla_freqs > 0, or Cobafa: cobafa_freqs) field with the Vanilla
    decoders."""
    n, R = packed.size(0), info.size(0)
    x = packed[:, :3]
    feat = features(sd, x, vanilla_freqs, cobafa_freqs)
    sig = _TruncExp.apply(mlp(sd, "sigma_decoder.net.net.", feat) - 1.).ravel()
    w = _Weights.apply(sig, packed[:, 6].contiguous(), info, thr)
    mask = w > 0
    rgbs = torch.zeros((n, 3))
    if mask.any():
        d = packed[:, 3:6][mask]
        inp = torch.cat([posenc(d, sd["rgb_decoder.pe.freqs"]), d, feat[mask]], -1)
        rgbs = rgbs.index_put((torch.nonzero(mask).squeeze(1),), torch.sigmoid(mlp(sd, "rgb_decoder.net.net.", inp)))
        rgbs = rgbs * w[:, None]
    else:                                      # core.py:251-254 "Empty iteration": fresh leaves, no parameter is reached
        rgbs = torch.zeros((n, 3), requires_grad=True)
        w = torch.zeros(n, requires_grad=True)
    idx = torch.repeat_interleave(torch.arange(R), info[:, 1].long())
    out = torch.zeros((R, 3)).index_add(0, idx, rgbs)
    if bg is not None:
        op = torch.zeros(R).index_add(0, idx, w)
        out = out + bg * (1 - op[:, None])
    return out


def loss_tv(sd, prefix: str = "feature_module.") -> torch.Tensor:
    """models.py:115-118,165-172."""
    vals = []
    for k, p in sd.items():
        if k.startswith(prefix) and k.endswith(".plane"):
            vals.append(torch.nn.functional.mse_loss(p[:, :, 1:, :], p[:, :, :-1, :]) +
                        torch.nn.functional.mse_loss(p[:, :, :, 1:], p[:, :, :, :-1]))
    return sum(vals) / len(vals)


def training_loss(sd, packed, info, target, bg, tv_alpha: float = 1e-4) -> torch.Tensor:
    """run.py:251-256 for method == kplanes."""
    out = render(sd, packed, info, bg)
    return torch.nn.functional.mse_loss(out, target) + tv_alpha * loss_tv(sd)


def weights_conditioning(compute_grads) -> Dict[str, float]:
    """How well the reference's fp32 weights backward determines each gradient tensor, relative to max |grads|: the largest
    deviation over (a) an exact evaluation of the same formula and (b) three draws of its rounding noise (every suffix sum of
    cuda.cu:49-56 is -total + prefix in fp32: an absolute error ~2^-23 sum |w g| of the ray, times the step size).  One draw is
    what any single implementation -- the reference on some host, the HIP kernels -- realises; tests allow 4 x this."""
    base = compute_grads()
    worst = {k: 0.0 for k in base}

    def fold(alt):
        for k in base:
            worst[k] = max(worst[k], float(np.abs(alt[k] - base[k]).max() / max(float(np.abs(base[k]).max()), 1e-30)))
    _Weights.exact_backward = True
    try:
        fold(compute_grads())
    finally:
        _Weights.exact_backward = False
    for seed in (1, 2, 3):
        _Weights.noise_probe = seed
        try:
            fold(compute_grads())
        finally:
            _Weights.noise_probe = None
    return worst


def grads_of(sd: Dict[str, torch.Tensor], loss_fn) -> Dict[str, np.ndarray]:
    """Run loss_fn on a leaf copy of every floating parameter and return d loss / d param."""
    leaves = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("freqs") else v)
              for k, v in sd.items()}
    loss = loss_fn(leaves)
    loss.backward()
    return {k: v.grad.numpy() for k, v in leaves.items() if isinstance(v, torch.Tensor) and v.grad is not None}, float(loss)


def initial_state(method: str, seed: int, kplanes_resolutions=(128, 256, 512)) -> Dict[str, torch.Tensor]:
    """The ``state_dict`` the reference's constructors leave behind ``torch.manual_seed(seed)`` (train.py:68-72, run.py:130-152): the
    same ``torch.nn`` initialisers drawing from the global CPU generator in the same order -- feature module (models.py:59-68 /
    123-151 / 234-250), ``VanillaOpacityDecoder(dim)`` (models.py:70-77), ``VanillaColorDecoder(8, dim, 64, 3)`` (models.py:79-89).
    Pinned bit for bit by the sha256 of every initial tensor of the reference's own ``train()`` (golden G22,
    tests/test_oracle_train_trace.py); the caller's RNG stream is left untouched."""
    sd: Dict[str, torch.Tensor] = {}

    def mlp_(prefix, n_in, hidden, n_hidden, n_out=None):             # models.py:7-26: Linear, ReLU, n x Sequential(Linear, ReLU), Linear
        n_out = hidden if n_out is None else n_out
        dims = [(prefix + "0", n_in, hidden)] + [(prefix + f"{2 + i}.0", hidden, hidden) for i in range(n_hidden)] + \
               [(prefix + f"{2 + n_hidden}", hidden, n_out)]
        for name, i, o in dims:
            lin = torch.nn.Linear(i, o)
            sd[name + ".weight"], sd[name + ".bias"] = lin.weight.detach(), lin.bias.detach()

    def freqs_(n):                                                    # models.py:33
        return 2 ** torch.arange(0, n) * torch.pi
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        if method == "vanilla":
            sd["feature_module.encoding.freqs"] = freqs_(10)
            mlp_("feature_module.net.net.", 60, 256, 8)
            dim = 256
        elif method == "kplanes":
            for s, r in enumerate(kplanes_resolutions):
                for p in range(3):
                    sd[f"feature_module.planes.{s}.{p}.plane"] = torch.nn.init.uniform_(torch.empty(1, 32, r, r))
            dim = 32 * len(kplanes_resolutions)
        elif method == "cobafa":
            for i, (r, c) in enumerate(zip(torch.linspace(32., 128, 6).int().tolist(), [8, 8, 8, 4, 4, 4])):
                sd[f"feature_module.basis_grids.{i}.grid"] = torch.nn.init.uniform_(torch.empty(1, c, r, r, r))
            sd["feature_module.coef_grid.grid"] = torch.nn.init.uniform_(torch.empty(1, 6, 64, 64, 64))
            mlp_("feature_module.mlp.net.", 36, 128, 5)
            dim = 128
        else:
            raise NotImplementedError(method)
        mlp_("sigma_decoder.net.net.", dim, 64, 0, 1)
        sd["rgb_decoder.pe.freqs"] = freqs_(8)
        mlp_("rgb_decoder.net.net.", dim + 8 * 2 * 3 + 3, 64, 3, 3)
    return sd


def reference_training(sd0: Dict[str, torch.Tensor], rays_o: np.ndarray, rays_d: np.ndarray, rgbs: np.ndarray, *,
                       method: str, batch_size: int, n_samples: int, n_steps: int, occupancy_res: int = 128,
                       bg=(1.0, 1.0, 1.0), grad_scale: float = 1024.0, vanilla_freqs: int = 10, scene_type: str = "aabb",
                       scene_scale: float = 1.0, cobafa_freqs=None, occ_updates: int = 0, grid0=None, grids_out=None,
                       stochastic_seed: Optional[int] = None, eval_at=(), eval_fn=None, on_step=None, replay: Optional[dict] = None,
                       lr: float = 1e-2, loader: str = "stream", lrs_out=None):
    """The reference's train() loop (run.py:97-319) on CPU in deterministic form: consecutive rays instead of a
    shuffled loader, no sampling jitter, voxel-centre occupancy refresh.  Literals as in run.py:100-114,186-202,
    including the scaled-and-never-unscaled loss.  Returns (losses, final state dict, per-step sample counts).
    Test knobs (defaults = the reference): ``occ_updates`` overrides the refresh period 16 * 4096 / B (run.py:103), ``grid0``
    the all-ones initial grid (core.py:108), ``grids_out`` (a list) receives (step, grid, mean) after every refresh.
    ``stochastic_seed`` (round 4): the loop AS THE REFERENCE RUNS IT -- a ``DataLoader(shuffle=True)`` stream of loader batches
    that persists across steps (a fresh permutation per epoch, the partial last batch included: run.py:116-122,221-225), sampling
    jitter ``t += U[0,1) * delta`` per candidate (core.py:172-173) and jittered voxel coordinates in the refresh (core.py:136), all
    from one numpy generator.  ``eval_fn(step, sd, grid, threshold)`` is called after ``step`` optimizer steps for every step in
    ``eval_at`` (the PSNR@step half of the metric, run.py:53-54).  ``on_step(step, sd, packed, info, target)`` is called with the
    parameters and the dynamic batch of every step right before its forward pass (gradient parity ALONG the reference's trajectory).
    ``replay = {"seed": s, "rank": r}``: stochastic like ``stochastic_seed`` but every random choice is taken from the streams the HIP
    harness defines for ``TrainConfig(seed=s, host_shuffle=True)`` (tinynerf_amd/run.py), which stand in for torch's global RNG there:
    ray order = successive ``torch.randperm(M, generator=Generator().manual_seed(s * 1000003 + r + 1), dtype=int32)`` walked in
    loader batches (an epoch's unread tail is continued by the next permutation -- the harness' documented deviation from the partial
    last batch of a DataLoader); sampling jitter of the step's b-th loader batch, ray i, candidate k = ``orc.uniform01(jitter_seed(s,
    step, r), (b * B + i) * S + k)``; voxel jitter of a refresh = ``orc.uniform01(refresh_seed(s, step), ((slice * H * W + h * W + w) * 3
    + c)``.  With the same parameters the two sides then walk the SAME rays with the SAME jitter: their trajectories differ by fp32
    summation order only (tests/test_hip_psnr.py, golden G18).
    ``loader="dataloader"`` (round 6, with ``replay``): the permutations are walked as ``DataLoader(shuffle=True)`` walks them
    (run.py:116-122,221-225) -- the partial last batch of an epoch is handed out as it is and the next epoch starts with a fresh
    permutation -- which is how ``oracle/make_train_trace.py`` drives the reference's own ``train()``: golden G22 holds this function
    to that run (tests/test_oracle_train_trace.py).  ``lrs_out`` (a list) receives the learning rate after every ``scheduler.step()``."""
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("freqs") else v.clone())
          for k, v in sd0.items()}
    params = [v for v in sd.values() if v.requires_grad]
    bs_ratio = 4096 / batch_size
    steps = int(2048 * bs_ratio)
    occ_updates = occ_updates or int(16 * bs_ratio)
    opt = torch.optim.Adam(params, lr=lr, eps=1e-15, weight_decay=1e-5)          # (lr: test knob; run.py:110 has 1e-2)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[steps // 2, steps * 3 // 4, steps * 5 // 6, steps * 9 // 10], gamma=0.33)
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    # run.py:154-162: aabb scenes march the box, unbounded ones the Mip-NeRF-360 table with the inf-norm contraction
    step_size = float(orc.aabb_step_size(aabb, n_samples)) if scene_type == "aabb" else scene_scale / n_samples
    decay = 0.01 ** (1 / 16)
    grid = np.ones((occupancy_res,) * 3, np.float32) if grid0 is None else np.array(grid0, np.float32, copy=True)
    mean = float(orc.torch_like_mean(grid))
    bg_t = None if bg is None else torch.tensor(bg, dtype=torch.float32)
    vf = vanilla_freqs if method == "vanilla" else 0
    cf = tuple(cobafa_freqs) if method == "cobafa" else None
    cursor, M = 0, rays_o.shape[0]
    target_size = batch_size * n_samples
    losses, counts = [], []
    rng = None if stochastic_seed is None else np.random.default_rng(stochastic_seed)
    eval_at = set(int(e) for e in eval_at)

    def shuffled_loader():              # DataLoader(shuffle=True) without drop_last, restarted on StopIteration (run.py:221-225)
        while True:
            perm = rng.permutation(M)
            for b0 in range(0, M, batch_size):
                idx = perm[b0:b0 + batch_size]
                yield rays_o[idx], rays_d[idx], rgbs[idx]
    stream = shuffled_loader() if rng is not None else None
    if replay is not None:
        assert rng is None, "replay and stochastic_seed exclude each other"
        import hashlib
        r_seed, r_rank = int(replay["seed"]), int(replay.get("rank", 0))
        host_gen = torch.Generator().manual_seed(r_seed * 1000003 + r_rank + 1)

        def replay_loader():            # tinynerf_amd/run.py Trainer._epoch_block with cfg.host_shuffle
            buf = np.empty(0, np.int64)
            while loader == "dataloader":
                perm = torch.randperm(M, generator=host_gen, dtype=torch.int32).numpy().astype(np.int64)
                for b0 in range(0, M, batch_size):
                    idx = perm[b0:b0 + batch_size]
                    yield rays_o[idx], rays_d[idx], rgbs[idx]
            while True:
                while buf.shape[0] < batch_size:
                    buf = np.concatenate([buf, torch.randperm(M, generator=host_gen, dtype=torch.int32).numpy().astype(np.int64)])
                idx, buf = buf[:batch_size], buf[batch_size:]
                yield rays_o[idx], rays_d[idx], rgbs[idx]
        stream = replay_loader()

        def jitter_seed(step):          # tinynerf_amd/run.py jitter_seed
            h = int.from_bytes(hashlib.blake2b(b"tinynerf-jitter:%d:%d" % (r_seed, step), digest_size=8).digest(), "little")
            return (h & (2 ** 62 - 1)) * 2 + 1 + r_rank

    def sigma_np(pts: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(pts))
            feat = features(sd, x, vf, cf)
            return torch.exp(mlp(sd, "sigma_decoder.net.net.", feat) - 1.).numpy()

    for step in range(n_steps):
        thr = min(0.01, mean)
        if step in eval_at and eval_fn is not None:
            eval_fn(step, sd, grid, thr)
        # dynamic batch (run.py:215-244) over consecutive loader batches
        def batches():
            c = cursor
            while True:
                idx = (c + np.arange(batch_size)) % M
                yield rays_o[idx], rays_d[idx], rgbs[idx]
                c += batch_size
        jit_of = (lambda o: None) if rng is None else (lambda o: rng.random((o.shape[0], n_samples), dtype=np.float32))
        if replay is not None:
            calls = []

            def jit_of(o, calls=calls, seed=jitter_seed(step)):          # b-th loader batch of the step: rays b * B .. of the block
                r0 = len(calls) * batch_size
                calls.append(r0)
                ctr = (np.arange(r0, r0 + o.shape[0], dtype=np.uint64)[:, None] * np.uint64(n_samples)
                       + np.arange(n_samples, dtype=np.uint64)[None, :])
                return orc.uniform01(seed, ctr)
        if scene_type == "aabb":
            prov = lambda o, d: orc.ray_provider(o, d, marcher="aabb", contraction="aabb", grid=grid, threshold=thr,
                                                 n_samples=n_samples, near=0.1, aabb=aabb, jitter=jit_of(o))
        else:
            prov = lambda o, d: orc.ray_provider(o, d, marcher="unbounded", contraction="mip360", grid=grid, threshold=thr,
                                                 n_samples=n_samples, near=0.1, far=1e5, uniform_range=scene_scale, order=float("inf"),
                                                 jitter=jit_of(o))
        packed, info, target, k = orc.dynamic_batch(batches() if stream is None else stream, prov, target_size)
        cursor = (cursor + k * batch_size) % M
        if step % occ_updates == 0:
            if replay is not None:
                rs = (r_seed * 7919 + 104729 * (step + 1)) % (2 ** 62)      # tinynerf_amd/run.py refresh_seed
                per = occupancy_res * occupancy_res * 3
                jit = [orc.uniform01(rs, np.uint64(i * per) + np.arange(per, dtype=np.uint64)).reshape(occupancy_res, occupancy_res, 3)
                       for i in range(occupancy_res)]
            elif rng is None:
                jit = [np.full((occupancy_res, occupancy_res, 3), 0.5, np.float32)] * occupancy_res
            else:
                jit = [rng.random((occupancy_res, occupancy_res, 3), dtype=np.float32) for _ in range(occupancy_res)]
            grid, mean = orc.occupancy_update(grid, sigma_np, step_size, 0.01, decay, mean, jit)
            if grids_out is not None:
                grids_out.append((step, grid.copy(), mean))
        if on_step is not None:
            on_step(step, sd, packed, info, target)
        out = render(sd, torch.from_numpy(packed), torch.from_numpy(info), bg_t, vanilla_freqs=vf, cobafa_freqs=cf)
        loss = torch.nn.functional.mse_loss(out, torch.from_numpy(target))
        if method == "kplanes":
            loss = loss + 1e-4 * loss_tv(sd)
        opt.zero_grad()
        (loss * grad_scale).backward()
        opt.step()
        sched.step()
        if lrs_out is not None:
            lrs_out.append(float(opt.param_groups[0]["lr"]))
        losses.append(float(loss.detach()))
        counts.append((int(packed.shape[0]), int(info.shape[0])))
    if n_steps in eval_at and eval_fn is not None:
        eval_fn(n_steps, sd, grid, min(0.01, mean))
    return losses, {k: v.detach() for k, v in sd.items()}, counts
