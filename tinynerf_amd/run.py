"""Training / evaluation harness -- the caller side of the hot path (reference ``src/run.py``).

Keeps the reference's recipe literal for literal (run.py:100-114,186-202: 2048*4096/B steps, occupancy
refresh every 16*4096/B steps, threshold .01, 128^3 grid, decay .01^(1/16), Adam(lr 1e-2, eps 1e-15,
wd 1e-5), MultiStepLR(1/2, 3/4, 5/6, 9/10; gamma .33), MSE + 1e-4 TV for K-Planes, and the GradScaler
quirk: the loss is scaled by 2^10 and never unscaled before ``optimizer.step()``, run.py:259-260) but
re-plumbs the loop MI355X-first:

* rays live in HBM as flat tables; loader batches are device-side index draws (no DataLoader workers);
* the dynamic batch (run.py:215-244) is built by ONE pass of the sampler over a block of loader batches,
  the projection rule ``int(cur*(1+1/k)) >= B*S`` is evaluated on the device (``tn_batch_plan``) and a
  single 16-byte read-back per step replaces the reference's >= 3 host syncs per loader batch;
* rays shard across ranks (one process per GPU); gradients are summed with RCCL all-reduce, the MSE is
  normalised by the GLOBAL ray count so the result equals the single-GPU loss.
"""
from __future__ import annotations

import os
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .core import (ContractionAABB, ContractionMip360, NerfRenderer, OccupancyGrid, RayMarcherAABB,
                   RayMarcherUnbounded, RayProvider)
from .optim import FusedAdam
from .models import (CobafaFeatureField, KPlanesFeatureField, VanillaColorDecoder, VanillaFeatureMLP,
                     VanillaOpacityDecoder)


def psnr(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """-10 log10(mse) -- run.py:53-54."""
    return -10. * torch.log10(torch.mean((x - y) ** 2))


from .config import CONFIG as _CONFIG      # noqa: E402
ADAM_OVERLAP = _CONFIG.adam_overlap      # TN_ADAM_OVERLAP; N == 1: the planes' optimizer pass beside the weight-gradient kernels (_planes_adam_early)
SIDE_PLAN = _CONFIG.side_plan            # TN_SIDE_PLAN; the next step's sampler pass on a stream of its own


@dataclass
class TrainConfig:
    method: str = "kplanes"            # vanilla | kplanes | cobafa   (run.py:130-152)
    scene_type: str = "aabb"           # aabb | unbounded             (run.py:154-162)
    batch_size: int = 1024             # rays per loader batch
    n_samples: int = 1024              # candidates per ray
    scene_scale: float = 1.0           # uniform_range of the unbounded marcher
    grad_scale: float = 2.0 ** 10      # GradScaler(2**10) without unscale (run.py:201,259-260)
    seed: int = 0
    occupancy_res: int = 128           # run.py:106
    deterministic: bool = False        # parity runs: consecutive rays, no sampling jitter, voxel-centre occupancy refresh
    kplanes_resolutions: Tuple[int, ...] = (128, 256, 512)      # models.py:126-142
    # strong scaling (N > 1): the recipe's batch is SPLIT over `shard` ranks -- each rank draws loader batches of batch_size / shard
    # rays and stops at batch_size * n_samples / shard packed samples, so a step of the job processes the recipe's B * S samples
    # whatever N is and "PSNR at equal step count" keeps its meaning; the schedule (steps, refresh period, LR milestones) follows
    # the GLOBAL batch_size.  shard = 1 with N > 1 is weak scaling: every rank runs the recipe's batch (N x the samples per step)
    shard: int = 1
    # the shuffled ray stream from the HOST generator (torch.randperm on the CPU, uploaded once per epoch) instead of the device's:
    # a stream that any machine can regenerate (the CPU checker's replay mode walks the same rays) at the
    # price of one host permutation + upload per epoch (fine for parity runs, not for the 12.8 M-ray tables of the bench)
    host_shuffle: bool = False
    # N > 1: how the plane gradients travel.  False: in-place all-reduce of the live rows, every rank runs Adam + TV on all 33 M plane
    # elements.  True (round 5): reduce-scatter -> Adam + TV on the rank's own 1 / N of every plane's rows -> all-gather of the updated
    # rows: half the bytes of an all-reduce on the wire before the optimizer and 1 / N of its 0.22 ms; the all-gather moves every row
    # (weight decay and the regulariser change dead rows too).  None: on when the recipe's batch is split over >= 4 ranks (shard >= 4:
    # DESIGN 5.1 -- below ~1.1 ms of compute per step the replicated optimizer pass is what limits strong scaling).
    sharded_optimizer: Optional[bool] = None


def jitter_seed(seed: int, batch_no: int, rank: int = 0) -> int:
    """Seed of the sampler's counter RNG (csrc/tn_common.h tn::uniform01) for the ``batch_no``-th dynamic batch of a run: a pure
    function of (TrainConfig.seed, batch number, rank) -- not of how often a candidate block had to be redrawn or of whether the
    pass was prefetched -- so that a run's sampling jitter can be regenerated anywhere (the tests' CPU restatement of the counter RNG does)."""
    import hashlib
    h = int.from_bytes(hashlib.blake2b(b"tinynerf-jitter:%d:%d" % (seed, batch_no), digest_size=8).digest(), "little")
    return (h & (2 ** 62 - 1)) * 2 + 1 + rank


def refresh_seed(seed: int, train_step: int) -> int:
    """Seed of the voxel jitter of the occupancy refresh at ``train_step`` (the same on every rank: identical grids)"""
    return (seed * 7919 + 104729 * (train_step + 1)) % (2 ** 62)


def build_renderer(cfg: TrainConfig, bg_color: Optional[torch.Tensor], device: torch.device):
    """Model / scene construction of run.py:124-182 with the reference's literals."""
    if cfg.method == "vanilla":
        feature_module: torch.nn.Module = VanillaFeatureMLP(10, 256, 8)
    elif cfg.method == "kplanes":
        feature_module = KPlanesFeatureField(32, cfg.kplanes_resolutions)
    elif cfg.method == "cobafa":
        feature_module = CobafaFeatureField(
            basis_res=torch.linspace(32., 128, 6).int().tolist(), coef_res=64,
            freqs=torch.linspace(2., 8., 6).tolist(), channels=[8, 8, 8, 4, 4, 4], mlp_hidden_dim=128)
    else:
        raise NotImplementedError(f"Unknown method {cfg.method}.")
    dim = feature_module.feature_dim
    sigma_decoder = VanillaOpacityDecoder(dim)
    rgb_decoder = VanillaColorDecoder(8, dim, 64, 3)
    if cfg.scene_type == "unbounded":
        ray_marcher = RayMarcherUnbounded(cfg.n_samples, 0.1, 1e5, uniform_range=cfg.scene_scale)
        contraction = ContractionMip360(order=float("inf"))
    elif cfg.scene_type == "aabb":
        aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]).to(device)
        ray_marcher = RayMarcherAABB(aabb, cfg.n_samples, 0.1)
        contraction = ContractionAABB(aabb)
    else:
        raise NotImplementedError(f"Unknown scene type {cfg.scene_type}.")
    occupancy_grid = OccupancyGrid(size=cfg.occupancy_res, step_size=ray_marcher.step_size, threshold=0.01,
                                   decay=0.01 ** (1 / 16)).to(device)
    ray_provider = RayProvider(occupancy_grid=occupancy_grid, contraction=contraction, ray_marcher=ray_marcher)
    renderer = NerfRenderer(feature_module, sigma_decoder, rgb_decoder, bg_color=bg_color).to(device)
    return renderer, occupancy_grid, ray_provider


class Trainer:
    """One object = the body of ``train()`` (run.py:97-319) with ``step()`` as the loop iteration."""

    def __init__(self, cfg: TrainConfig, rays_o: torch.Tensor, rays_d: torch.Tensor, rgbs: torch.Tensor,
                 bg_color: Optional[torch.Tensor], device: torch.device, rank: int = 0, world_size: int = 1):
        self.cfg, self.device, self.rank, self.world = cfg, device, rank, world_size
        self.rays_o, self.rays_d, self.rgbs = rays_o, rays_d, rgbs
        with torch.random.fork_rng(devices=[]):         # identical parameters on every rank, the caller's RNG stream untouched
            torch.manual_seed(cfg.seed)
            self.renderer, self.occupancy_grid, self.ray_provider = build_renderer(cfg, bg_color, device)
        bs_ratio = 4096 / cfg.batch_size
        self.steps = int(2048 * bs_ratio)
        self.occupancy_grid_updates = int(16 * bs_ratio)
        self.tv_reg_alpha, self.l1_reg_alpha = 0.0001, 0.
        if cfg.shard < 1 or cfg.batch_size % cfg.shard:
            raise ValueError(f"TrainConfig.shard = {cfg.shard} must divide batch_size = {cfg.batch_size}")
        self.loader_batch = cfg.batch_size // cfg.shard              # rays per loader batch ON THIS RANK
        self.target_sample_size = self.loader_batch * cfg.n_samples
        params = list(self.renderer.parameters())
        for p in params:                                # grads keep the parameter's (channels_last) layout
            p.grad = torch.zeros_like(p)
        # N > 1: every small gradient is a view into ONE flat buffer, all-reduced in place as a single bucket (no cat / copy
        # kernels); its last slot carries the step's "Empty iteration" gate through the same collective
        self._flat, self._flat_used, self._flat_ids = Trainer._flat_small_grads(params, device) if world_size > 1 else (None, 0, set())
        self.renderer.accumulate_into_grad = True       # fused path adds into these buffers directly
        self.renderer.reuse_buffers = True              # and keeps its scratch in a capacity-based arena
        # ... as do the field's own MLP stacks (Vanilla 256 x 10: 10 KB of activation workspace per sample = 11 GB per step;
        # Cobafa 128 x 6), sized once from the dynamic batch's target so that no step ever meets hipMalloc
        from .arena import Arena
        from .models import MLP
        self.scratch = Arena()
        for i, m in enumerate(mod for mod in self.renderer.feature_module.modules() if isinstance(mod, MLP)):
            m.__dict__["scratch"] = (self.scratch, f"mlp_ws{i}", {}, True, {})      # arena, name, row-view link, -, persistent state
        if isinstance(self.renderer.feature_module, CobafaFeatureField):
            self.renderer.feature_module.__dict__["accumulate_into_grad"] = True
        self._arena: Dict[str, torch.Tensor] = {}
        self._arena_grown = 0
        # torch.optim.Adam's update (run.py:186), one kernel pass per tensor, gradients zeroed in the same pass
        self.optimizer = FusedAdam(params, lr=1e-2, eps=1e-15, weight_decay=1e-5, zero_grad_in_step=True)
        # the planes' moments and double buffer exist before the first step, allocated HERE on the main stream: the overlapped pass
        # (TN_ADAM_OVERLAP) runs on a side stream, and blocks first allocated there would belong to that stream's pool although they
        # become `plane.data` and are read on the main stream ever after
        if isinstance(self.renderer.feature_module, KPlanesFeatureField):
            self.optimizer.preallocate(self.renderer.feature_module.plane_tensors(), shadow=True)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(
            self.optimizer, milestones=[self.steps // 2, self.steps * 3 // 4, self.steps * 5 // 6, self.steps * 9 // 10],
            gamma=0.33)
        self.train_step = 0
        self._cursor = 0                 # deterministic mode: position in the ray table
        self._perm: Optional[torch.Tensor] = None      # random mode: the current shuffled epoch of ray indices (int32) ...
        self._perm_pos = 0                             # ... the position of the next loader batch in it ...
        self._carry = torch.empty(0, dtype=torch.int32, device=device)     # ... and the unread tail of the previous epoch
        self._k_guess = 8
        # per-rank ray stream over the rank's own ray table (see _epoch_block): same generator family, different seed
        self._gen = torch.Generator(device=device)
        self._gen.manual_seed(cfg.seed * 1000003 + rank + 1)
        self._host_gen = torch.Generator().manual_seed(cfg.seed * 1000003 + rank + 1)       # cfg.host_shuffle
        self._batch_no = 0               # dynamic batches handed out so far (seeds the sampling jitter: jitter_seed)
        self.last: Dict[str, float] = {}
        self.grad_hook: Optional[Callable[["Trainer"], None]] = None   # called with the final (reduced) gradients, before Adam
        self._early: Dict[int, object] = {}              # all-reduces started during the backward pass (N > 1)
        self._pending: Optional[dict] = None             # sampler pass of the next step, already in flight
        # N > 1, K-Planes: rows of every plane that can receive a gradient at all (see _refresh_reduce_rows)
        self._plane_of: Dict[int, int] = {}
        self._reduce_rows: Optional[List[Tuple[int, int]]] = None
        self._reduce_rows_prev: Optional[List[Tuple[int, int]]] = None
        if world_size > 1 and isinstance(self.renderer.feature_module, KPlanesFeatureField):
            self._plane_of = {id(p): i for i, p in enumerate(self.renderer.feature_module.plane_tensors())}
        self._sharded = bool(cfg.sharded_optimizer and world_size > 1 and self._plane_of and
                             all(p.size(2) % world_size == 0 for p in self.renderer.feature_module.plane_tensors()))
        self._gathers: List[object] = []
        if self._sharded:
            self.optimizer.partial_state_reason = Trainer._PARTIAL
        self._plan_host: Optional[torch.Tensor] = None
        self._info_turn = 0
        self._side: Optional[torch.cuda.Stream] = None
        self._side2: Optional[torch.cuda.Stream] = None          # TN_ADAM_OVERLAP: the planes' optimizer pass
        self._gate_ring = torch.zeros(256, device=device)
        self._gate_tick = 0
        self._acc_ring = torch.zeros((64, 1 + 32 * 3), dtype=torch.float64, device=device)
        self._acc_tick = 0               # a counter of its own (train_step may be assigned from outside, a step may be retried)
        self.prefetch = True
        # tensors that a sampler pass in flight on the side stream may still read although the trainer has dropped them (the
        # previous epoch's permutation, the carry it was cut from, an arena buffer that was just regrown): kept until the
        # pass's event has been synchronised (build_batch) -- the caching allocator would otherwise hand their blocks to
        # whichever stream allocated them while the other stream's kernel is still queued
        self._graveyard: List[torch.Tensor] = []

    def __del__(self):
        side = getattr(self, "_side", None)
        if side is not None:             # a side pass may still be writing into buffers this object owns
            try:
                side.synchronize()
            except Exception:            # noqa: BLE001 -- interpreter shutdown
                pass

    def _on_stream(self, *tensors: Optional[torch.Tensor]) -> None:
        """mark tensors as in use on the CURRENT stream (no-op for the stream they were allocated on): the sampler pass runs on
        the main stream in step 0 / in a redraw and on the side stream otherwise, and its tensors cross over"""
        cur = torch.cuda.current_stream(self.device)
        for t in tensors:
            if t is not None and t.is_cuda and t.numel():
                t.record_stream(cur)

    def _buf(self, name: str, shape, dtype) -> torch.Tensor:
        """capacity-based scratch (see fused.Arena): sizes drift by a few percent per step"""
        numel = 1
        for d in shape:
            numel *= int(d)
        t = self._arena.get(name)
        if t is None or t.numel() < numel:
            if t is not None:
                self._graveyard.append(t)
            t = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=self.device)
            self._arena[name] = t
            self._arena_grown += 1
        self._on_stream(t)
        return t[:numel].view(*shape)

    # ------------------------------------------------------------------ a8: dynamic batch
    def _epoch_block(self, m: int) -> torch.Tensor:
        """The next ``m`` ray indices of the shuffled stream: the reference's ``DataLoader(shuffle=True)`` (run.py:116-122) walks a
        fresh permutation of all rays per epoch -- every ray once per epoch, no replacement -- in loader batches of B; here the
        permutation lives on the device (int32 ``torch.randperm`` from the trainer's own generator) and a step consumes the k
        batches its dynamic batch took (``_advance`` in build_batch; a block that is redrawn larger starts at the same place).
        The stream is  carry ++ perm[pos:] : at an epoch boundary only the few unread indices of the old permutation are kept
        (``_carry``), the new permutation is never concatenated or copied.  N > 1: every rank is handed its OWN share of the
        rays (bench.py: its own views; the tests: ``rays[rank::world]``), so per-rank permutations of per-rank tables are
        disjoint by construction and an epoch of the job visits every ray once.  Deviation: the reference's partial last
        loader batch of an epoch (DataLoader without drop_last) is filled up from the next epoch instead."""
        n_rays = self.rays_o.size(0)
        self._on_stream(self._perm, self._carry)
        while self._carry.numel() + (0 if self._perm is None else self._perm.numel() - self._perm_pos) < m:
            if self._perm is not None:
                self._graveyard += [self._perm, self._carry]        # still read by the `cat` below, possibly on the other stream
                self._carry = torch.cat([self._carry, self._perm[self._perm_pos:]])
            if self.cfg.host_shuffle:
                self._perm = torch.randperm(n_rays, generator=self._host_gen, dtype=torch.int32).to(self.device)
            else:
                self._perm = torch.randperm(n_rays, device=self.device, generator=self._gen, dtype=torch.int32)
            self._perm_pos = 0
        c = self._carry.numel()
        if c == 0:
            return self._perm[self._perm_pos:self._perm_pos + m]
        return torch.cat([self._carry[:m], self._perm[self._perm_pos:self._perm_pos + max(0, m - c)]])

    def _advance(self, r: int) -> None:
        """consume the first ``r`` indices of the shuffled stream"""
        c = self._carry.numel()
        if r >= c:
            self._carry = self._carry[:0]
            self._perm_pos += r - c
        else:
            self._carry = self._carry[r:]

    @torch.no_grad()
    def _launch_plan(self, n_b: Optional[int] = None) -> None:
        """First half of the dynamic-batch rule (run.py:215-244): draw a block of rays, count their live samples
        (tn_sample_mask) and find the cut (tn_batch_plan).  The 16-byte plan goes to pinned host memory behind an
        event, so ``step()`` can enqueue this right after the forward pass: the numbers are on the host long before
        the backward pass has drained, and the host never waits for the whole queue (one read-back per step, but no
        pipeline bubble).  The draw depends on the occupancy grid and the ray stream only, not on the parameters."""
        cfg, dev = self.cfg, self.device
        B, S = self.loader_batch, cfg.n_samples
        n_chunks = (S + 63) // 64
        # candidate block: the rule trips after k loader batches and k drifts by a batch or two between steps; a block that
        # turns out too small is redrawn at twice the size (build_batch), so the margin only has to cover the drift
        n_b = n_b or min(4096, max(2, int(self._k_guess * 1.12) + 3))
        R_all = n_b * B
        if cfg.deterministic:
            idx = ((self._cursor + torch.arange(R_all, device=dev)) % self.rays_o.size(0)).to(torch.int32)
        else:
            idx = self._epoch_block(R_all)                     # int32
        # origins, directions and target colours of the candidate rays in one launch (three index kernels + an index cast
        # before); two buffers in turn: the previous step's backward pass still reads its own directions and targets
        self._info_turn ^= 1
        o = self._buf(f"cand_o{self._info_turn}", (R_all, 3), torch.float32)
        d = self._buf(f"cand_d{self._info_turn}", (R_all, 3), torch.float32)
        rgb = self._buf(f"cand_rgb{self._info_turn}", (R_all, 3), torch.float32)
        L.call("tn_gather_rays", dev, L.ptr(self.rays_o), L.ptr(self.rays_d), L.ptr(self.rgbs), L.ptr(idx), C.c_int64(R_all), L.ptr(o), L.ptr(d),
               L.ptr(rgb))
        desc = self.ray_provider._desc(dev, not cfg.deterministic, None)
        desc.seed = jitter_seed(cfg.seed, self._batch_no, self.rank)       # (a redrawn block repeats it: same rays, same jitter)
        maskbits = self._buf("maskbits", (R_all, n_chunks), torch.int64)
        counts = self._buf("counts", (R_all,), torch.int32)
        plan = self._buf("plan", (4,), torch.int32)
        # (offset, count) of EVERY candidate ray, in the same launch as the rule (the scan of a prefix does not depend on where the
        # rule cuts): behind the read-back only the pack is left.  Two buffers in turn: this runs during the previous step, whose
        # backward pass still reads its own info.
        info = self._buf(f"info{self._info_turn}", (R_all, 2), torch.int32)
        L.call("tn_sample_mask", dev, C.byref(desc), L.ptr(o), L.ptr(d), C.c_int64(R_all), L.ptr(maskbits), L.ptr(counts))
        L.call("tn_batch_plan_scan", dev, L.ptr(counts), C.c_int64(R_all), C.c_int32(B), C.c_int64(self.target_sample_size), L.ptr(plan),
               L.ptr(info))
        if self._plan_host is None:
            self._plan_host = torch.empty(4, dtype=torch.int32, pin_memory=True)
            self._plan_event = torch.cuda.Event()
        self._plan_host.copy_(plan, non_blocking=True)
        self._plan_event.record(torch.cuda.current_stream(dev))
        self._pending = dict(n_b=n_b, idx=idx, o=o, d=d, rgb=rgb, desc=desc, maskbits=maskbits, counts=counts, info=info)

    @torch.no_grad()
    def build_batch(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
        """packed [N,7], info [R,2], target rgbs [R,3], k -- run.py:215-244 in one sampler pass.
        packed / info are views of the trainer's reused scratch: valid until the next build_batch() (clone to keep)."""
        dev, B = self.device, self.loader_batch
        while True:
            if self._pending is None:
                self._launch_plan()
            pend, self._pending = self._pending, None
            self._plan_event.synchronize()              # the step's single host read-back
            del self._graveyard[:]                      # (everything the pass read has been read)
            self._on_stream(*(pend[k] for k in ("idx", "o", "d", "rgb", "maskbits", "counts", "info")))   # consumed on this stream now
            k, n, R, tripped = self._plan_host.tolist()
            if tripped or pend["n_b"] >= 4096:
                break
            self._k_guess = pend["n_b"] * 2             # not enough rays drawn: redraw a larger block
        self._k_guess = k
        self._batch_no += 1
        self.last_plan_seed = int(pend["desc"].seed)      # (the counter RNG's seed of this batch's sampling jitter: the tests read it)
        self._cursor = (self._cursor + R) % self.rays_o.size(0)
        if not self.cfg.deterministic:
            self._advance(R)
        info = pend["info"][:R]
        packed = self._buf("packed", (n, 7), torch.float32)
        ray_ids = self._buf("ray_ids", (n,), torch.int32)
        steps = self._buf("steps", (n,), torch.float32)
        L.call("tn_sample_pack", dev, C.byref(pend["desc"]), L.ptr(pend["o"]), L.ptr(pend["d"]), C.c_int64(R), L.ptr(pend["maskbits"]),
               L.ptr(info), C.c_void_p(None), L.ptr(packed), L.ptr(ray_ids), L.ptr(steps), C.c_int64(n))
        # what the fused render node would otherwise rebuild from (packed, info): ray id and step of every sample, ray directions
        # "Empty iteration" flag of this step: one slot of a ring that is zeroed once per lap (the weights kernel only raises it)
        slot = self._gate_tick % self._gate_ring.numel()          # (a counter of its own: a batch may be built without a step)
        self._gate_tick += 1
        if slot == 0 and self._gate_tick > 1:
            self._gate_ring.zero_()
        self.renderer._batch_aux = {"key": (packed.data_ptr(), n, R), "ray_ids": ray_ids, "steps": steps, "dirs": pend["d"][:R],
                                    "planes_ready": self._planes_ready if self.world > 1 else (self._planes_adam_early if ADAM_OVERLAP else None),
                                    "gate": self._gate_ring[slot:slot + 1]}
        return packed, info, pend["rgb"][:R], k

    # ------------------------------------------------------------------ one optimizer step
    def sigma_fn(self, t: torch.Tensor) -> torch.Tensor:
        return self.renderer.sigma_decoder(self.renderer.feature_module(t))

    def step(self) -> Dict[str, float]:
        packed, info, target, k = self.build_batch()
        return self.step_on_batch(packed, info, target, k)

    def step_on_batch(self, packed: torch.Tensor, info: torch.Tensor, target: torch.Tensor, k: int = 0,
                      prefetch: Optional[bool] = None) -> Dict[str, float]:
        """Everything of one optimizer step behind the dynamic batch (run.py:246-261) on an explicit batch."""
        cfg = self.cfg
        self.renderer.train()
        if self.train_step % self.occupancy_grid_updates == 0:                    # run.py:248-249
            jit = None
            if cfg.deterministic:
                r = cfg.occupancy_res
                jit = torch.full((r, r, r, 3), 0.5, device=self.device)
            # same jitter on every rank (identical grids without communication); a dedicated seed, not the global RNG
            self.occupancy_grid.update(self.sigma_fn, jitters=jit, seed=refresh_seed(cfg.seed, self.train_step))
            self._refresh_reduce_rows()
        ray_count = None
        if self.world > 1:                 # global ray count of the step (see global_ray_count): travels during the forward pass
            ray_count = torch.full((1,), float(info.size(0)), device=self.device)
            ray_count_done = torch.distributed.all_reduce(ray_count, async_op=True)
        rendered = self.renderer(packed, info)                                    # run.py:251
        # "Empty iteration" (core.py:251-254): no sample of the step has w > 0 -> no parameter is reached by the image loss, and
        # torch.optim.Adam skips the grad-is-None parameters (run.py:258-260); decided on the device through this scalar (the
        # step's largest weight, the tensor the render node's backward reads).  N > 1: the step on the union of all ranks' rays
        # is empty only when every rank's is, so the scalar is summed over ranks BEFORE the backward pass (a rank whose own batch
        # is fully masked must still propagate d / d sigma and the background term); 4 bytes, in flight behind the sampler
        # pass of the next step
        gate = getattr(self.renderer, "_stats", {}).get("gate")
        gate_done = None
        if self.world > 1 and gate is not None:
            gate_done = torch.distributed.all_reduce(gate, async_op=True)
        if self.prefetch if prefetch is None else prefetch:
            # next step's sampler pass (ray gather, occupancy masks, rule + scan, 16-byte read-back): it depends on the grid and the
            # ray stream only, so it goes to a stream of its own behind this forward pass -- VALU work that fills in beside the
            # HBM-bound weight-gradient kernels of the backward pass instead of standing in line with them.  build_batch() waits for
            # its event on the host before anything of the next step is launched, and its buffers are the other turn's.
            if not SIDE_PLAN:        # (same stream: debugging / A-B timing)
                self._launch_plan()
            else:
                if self._side is None:
                    self._side = torch.cuda.Stream(self.device)
                self._side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._side):
                    self._launch_plan()
        # loss * grad_scale, scaled and never unscaled (run.py:259-260 quirk).  The MSE and its gradient are written out
        # by hand (4 small kernels instead of ~12 through autograd); the regulariser's value and gradient are one launch.
        # (gradients were zeroed by the previous optimizer pass: zero_grad -> backward -> step, run.py:258-260)
        R = rendered.size(0)
        # [0]: sum of squares, [1:]: regulariser sums -- one row of a ring that is zeroed once per lap (no fill launch per step)
        row = self._acc_tick % self._acc_ring.size(0)
        self._acc_tick += 1
        if row == 0:
            self._acc_ring.zero_()
        acc = self._acc_ring[row]
        grad = self._buf("grad_rendered", (R, 3), torch.float32)
        if self.world == 1:
            inv, inv_dev = 1.0 / (3.0 * R), None
        else:                                                                     # MSE over ALL ranks' rays (see global_ray_count)
            ray_count_done.wait()
            if gate_done is not None:
                gate_done.wait()
            inv, inv_dev = 1.0, (1.0 / (3.0 * ray_count)).float()
        stats = getattr(self.renderer, "_stats", {})
        if gate is not None and stats.get("pre_gated"):
            # the render node raised the flag inside the weights kernel: the gradient leaves the loss kernel gated, and the node
            # is told so (it gates by itself otherwise)
            stats["upstream_gated"] = True
            L.call("tn_mse_grad_gated", self.device, L.ptr(rendered.detach()), L.ptr(target), C.c_int64(3 * R), C.c_float(2.0 * cfg.grad_scale * inv),
                   L.ptr(inv_dev), L.ptr(gate), L.ptr(grad), L.ptr(acc))
        else:
            L.call("tn_mse_grad", self.device, L.ptr(rendered.detach()), L.ptr(target), C.c_int64(3 * R), C.c_float(2.0 * cfg.grad_scale * inv),
                   L.ptr(inv_dev), L.ptr(grad), L.ptr(acc))
        self._early_adam = None
        if ADAM_OVERLAP and self.world == 1 and cfg.method == "kplanes" and self.grad_hook is None:
            spec0, _ = self.renderer.feature_module.regulariser_spec(self.tv_reg_alpha, self.l1_reg_alpha)   # type: ignore
            self._early_adam = {"plane_reg": {"spec": spec0, "upstream": cfg.grad_scale, "sums": acc[1:]}, "gate": gate, "done": False}
        early_done = False
        try:
            rendered.backward(grad)
            early_done = bool(self._early_adam is not None and self._early_adam.get("done"))
        finally:
            stats["upstream_gated"] = False
            if not early_done:
                # backward raised (or the hook never ran): nothing may be left armed for a later backward.  Note that a step whose backward
                # fails AFTER the hook ran has already moved the planes (TN_ADAM_OVERLAP): such a step is not retryable -- rebuild the Trainer
                # from a checkpoint instead (TN_ADAM_OVERLAP=0 keeps the whole update behind the backward pass)
                self._early_adam = None if self._early_adam is None or not self._early_adam.get("done") else self._early_adam
        reg_coef, plane_reg = None, None
        if cfg.method == "kplanes":                                               # run.py:254-256
            # the regulariser's gradient is the same on every rank (same planes): it is folded into the optimizer pass, after
            # the gradient exchange, with its full weight (grad_scale: the loss is scaled and never unscaled)
            spec, reg_coef = self.renderer.feature_module.regulariser_spec(self.tv_reg_alpha, self.l1_reg_alpha)   # type: ignore
            plane_reg = {"spec": spec, "upstream": cfg.grad_scale, "sums": acc[1:]}
            if self._sharded:                # this rank's rows of every plane: the only ones its optimizer pass touches
                plane_reg["rows"] = {id(p): Trainer._own_rows(p.size(2), self.rank, self.world) for p in self.renderer.feature_module.plane_tensors()}
        self._loss_parts = (acc, inv, inv_dev, reg_coef)
        if self.world > 1:
            self.all_reduce_grads(gate)
        if self.grad_hook is not None:
            self.grad_hook(self)
        if self._early_adam is not None and self._early_adam.get("done"):
            self.optimizer.step(plane_reg=plane_reg, gate=gate, only="rest")      # (the planes' pass is on its way on the side stream)
            torch.cuda.current_stream(self.device).wait_stream(self._side2)
        else:
            self.optimizer.step(plane_reg=plane_reg, gate=gate)
        if self._sharded:                    # the updated rows of every rank to every rank (all of them: weight decay and TV move dead rows too)
            works = []
            for p in self.renderer.feature_module.plane_tensors():
                works += Trainer._all_gather_rows(p.data, self.rank, self.world)
            for w in works:
                w.wait()
        self.scheduler.step()
        self.train_step += 1
        self.last = {"n_samples": float(packed.size(0)), "n_rays": float(info.size(0)), "k": float(k)}
        return self.last

    def global_ray_count(self, local_rays: int, gate: Optional[torch.Tensor] = None):
        """MSE over ALL ranks' rays: sum of local squared errors / (3 * global ray count).  Dynamic batches
        give every rank a different ray count, so a per-rank mean followed by gradient averaging would not
        equal the single-GPU loss; with this normalisation the SUM of rank gradients does.
        ``gate`` (the step's max weight, in place): replaced by the sum over ranks in the same exchange -- the step on the
        union of all ranks' rays is an "Empty iteration" only when every rank's is."""
        if self.world == 1:
            return float(local_rays)
        both = torch.full((2,), float(local_rays), device=self.device)     # a fill kernel: no pageable H2D copy, no host sync
        if gate is not None:
            both[1:2] = (gate > 0).float()
        torch.distributed.all_reduce(both)
        if gate is not None:
            gate.copy_(both[1:2])
        return both[0]

    def global_mse(self, rendered: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ((rendered - target) ** 2).sum() / (3.0 * Trainer.global_ray_count(self, rendered.size(0)))

    def loss_device(self) -> torch.Tensor:
        """Loss of the last step (MSE + weighted regulariser, run.py:252-256) as a 1-element device tensor, assembled
        from the accumulators the step left behind (valid until the next step); no host sync."""
        acc, inv, inv_dev, reg_coef = self._loss_parts
        v = acc[0] * inv * (inv_dev[0].double() if inv_dev is not None else 1.0)
        if reg_coef is not None:                  # (every rank holds the full sums -- or, sharded optimizer pass, its rows' share of them)
            v = v + (acc[1:1 + reg_coef.numel()] * reg_coef.reshape(-1)).sum() / (1 if getattr(self, "_sharded", False) else self.world)
        v = v.float().reshape(1).clone()
        if self.world > 1:
            torch.distributed.all_reduce(v)
        # a parameter that is not finite: the reference's next forward reports NaN (torch.relu hands NaN on, models.py:7-28) where the
        # kernels' v_max_f32 ReLU swallows it -- the optimizer pass raises a flag (planes: their regulariser sums above are NaN already)
        bad = self.optimizer.nonfinite_flag(self.device) if hasattr(self.optimizer, "nonfinite_flag") else None
        if bad is not None:
            v = torch.where(bad != 0, torch.full_like(v, float("nan")), v)
        return v

    def loss_value(self) -> float:
        return float(self.loss_device().item())

    # ------------------------------------------------------------------ e: gradient exchange
    @staticmethod
    def _dense_view(g: torch.Tensor) -> Optional[torch.Tensor]:
        """`g`'s memory as a contiguous tensor WITHOUT a copy, for in-place collectives: K-Planes gradients are
        channels_last [1,C,H,W], Cobafa grids channels_last_3d [1,C,D,H,W] (neither is `is_contiguous()`); None when the
        tensor is not one dense block in some dimension order."""
        if g.is_contiguous():
            return g
        order = sorted(range(g.dim()), key=lambda d: (-g.stride(d), -g.size(d)))
        v = g.permute(order)
        return v if v.is_contiguous() else None

    @torch.no_grad()
    def _refresh_reduce_rows(self) -> None:
        """N > 1, K-Planes: the image-loss gradient of a plane is exactly zero outside the rows its samples can touch, and
        samples only exist where the trilinear occupancy lookup passes -- next to a grid node above the threshold.  The grids
        are identical on all ranks (same parameters, same jitter seed), so every rank derives the same row range per plane
        from the node range along the plane's v axis (+- one node, +- one texel row) and the all-reduce moves only those rows
        (contiguous in the channel-last layout): for a scene that fills the middle half of the box, half of the 126 MiB.  The
        batch of the step that refreshes the grid was still sampled with the previous grid: the union of both ranges is used.
        One 6-integer read-back per refresh (every 16 * 4096 / B steps)."""
        if not self._plane_of:
            return
        g = self.occupancy_grid
        occ = g.grid > g.threshold                                   # [D, H, W] <-> (z, y, x)
        lo_hi = []
        for keep in (2, 1, 0):                                       # x, y, z: reduce over the other two dimensions
            line = occ.amax(dim=tuple(d for d in range(3) if d != keep)).to(torch.int32)
            idx = torch.arange(line.numel(), device=line.device, dtype=torch.int32)
            big = line.numel() + 1
            lo_hi += [torch.where(line > 0, idx, torch.full_like(idx, big)).amin(), torch.where(line > 0, idx, torch.full_like(idx, -1)).amax()]
        vals = torch.stack(lo_hi).tolist()
        dims = (g.grid.size(2), g.grid.size(1), g.grid.size(0))      # nodes along x, y, z
        rows: List[Tuple[int, int]] = []
        for i, p in enumerate(self.renderer.feature_module.plane_tensors()):
            axis = (1, 2, 2)[i % 3]                                  # v axis of plane (x,y), (x,z), (y,z): models.py:144-146
            i0, i1, S, Hp = vals[2 * axis], vals[2 * axis + 1], dims[axis], p.size(2)
            if i1 < 0:
                rows.append((0, 0))
                continue
            v_lo = max(-1.0, -1.0 + 2.0 * (i0 - 1) / max(S - 1, 1)); v_hi = min(1.0, -1.0 + 2.0 * (i1 + 1) / max(S - 1, 1))
            r0 = int(math.floor((v_lo + 1.0) * 0.5 * (Hp - 1))) - 1
            r1 = int(math.floor((v_hi + 1.0) * 0.5 * (Hp - 1))) + 3
            rows.append((max(0, r0), min(Hp, r1)))
        prev = self._reduce_rows if self._reduce_rows is not None else [(0, p.size(2)) for p in self.renderer.feature_module.plane_tensors()]
        self._reduce_rows_prev, self._reduce_rows = prev, rows

    def _reduce_view(self, g: torch.Tensor, plane: Optional[int]) -> Optional[torch.Tensor]:
        """the part of gradient `g` that has to travel: all of it, or -- for K-Planes plane number `plane` -- its live rows"""
        flat = Trainer._dense_view(g) if g.numel() >= (1 << 18) else None
        if flat is None or plane is None or self._reduce_rows is None or flat.dim() != 4 or not g.is_contiguous(memory_format=torch.channels_last):
            return flat
        (a0, a1), (b0, b1) = self._reduce_rows[plane], self._reduce_rows_prev[plane]
        r0, r1 = (min(a0, b0), max(a1, b1)) if a1 > a0 and b1 > b0 else ((a0, a1) if a1 > a0 else (b0, b1))
        return flat[:, r0:r1] if r1 > r0 else flat[:, :0]            # flat: [1, H, W, C] in memory order

    @staticmethod
    def _all_reduce_many(views):
        """In-place all-reduce of several dense tensors as ONE collective call where the backend has one (RCCL: one grouped launch
        instead of one launch -- and 5-20 us of host time beside an idle GPU -- per tensor); returns a waitable.  Public API only:
        torch.distributed.all_reduce_coalesced (round 4 used the private _coalescing_manager)."""
        views = [v for v in views if v.numel()]
        if not views:
            return None
        if len(views) == 1:
            return torch.distributed.all_reduce(views[0], async_op=True)
        if torch.distributed.get_backend() != "nccl" and views[0].is_cuda:      # (gloo's coalesced form takes CPU tensors only)
            works = [torch.distributed.all_reduce(v, async_op=True) for v in views]

            class _All:
                def wait(self_inner):
                    for w in works:
                        w.wait()
            return _All()
        coalesced = getattr(torch.distributed, "all_reduce_coalesced", None)
        if coalesced is None:                               # (a torch that has dropped the deprecated call: one async all-reduce per tensor)
            works = [torch.distributed.all_reduce(v, async_op=True) for v in views]

            class _Each:
                def wait(self_inner):
                    for w in works:
                        w.wait()
            return _Each()
        import warnings
        with warnings.catch_warnings():                     # (the call is public; torch announces its future replacement on every use -- only that)
            warnings.filterwarnings("ignore", message=".*all_reduce_coalesced.*", category=FutureWarning)
            warnings.filterwarnings("ignore", message=".*all_reduce_coalesced.*", category=UserWarning)
            warnings.filterwarnings("ignore", message=".*all_reduce_coalesced.*", category=DeprecationWarning)
            return coalesced(views, async_op=True)

    # ---- sharded optimizer pass (TrainConfig.sharded_optimizer): rank r owns rows [r H / N, (r + 1) H / N) of every plane ----
    @staticmethod
    def _own_rows(H: int, rank: int, world: int) -> Tuple[int, int]:
        return H * rank // world, H * (rank + 1) // world

    @staticmethod
    def _reduce_scatter_rows(g: torch.Tensor, rank: int, world: int):
        """sum of all ranks' gradient `g` (a dense channels_last plane gradient, H % world == 0) into THIS rank's rows of its own
        buffer, in place; the other rows keep partial sums nobody reads (the optimizer pass clears them).  Returns a waitable."""
        flat = Trainer._dense_view(g).view(-1)
        chunk = flat.numel() // world
        own = flat[rank * chunk:(rank + 1) * chunk]
        if torch.distributed.get_backend() == "nccl" or not flat.is_cuda:
            return torch.distributed.reduce_scatter_tensor(own, flat, async_op=True)
        return torch.distributed.all_reduce(flat, async_op=True)      # (gloo on GPU tensors -- the shared-GPU tests: same sums in the rank's rows)

    @staticmethod
    def _all_gather_rows(p: torch.Tensor, rank: int, world: int):
        """every rank's own rows of `p` (updated by its optimizer pass) to every rank, in place; returns waitables"""
        flat = Trainer._dense_view(p).view(-1)
        chunk = flat.numel() // world
        if torch.distributed.get_backend() == "nccl" or not flat.is_cuda:
            return [torch.distributed.all_gather_into_tensor(flat, flat[rank * chunk:(rank + 1) * chunk], async_op=True)]
        return [torch.distributed.broadcast(flat[r * chunk:(r + 1) * chunk], src=r, async_op=True) for r in range(world)]

    def optimizer_state_dict(self) -> dict:
        """``optimizer.state_dict()`` that is complete on every rank: under ``sharded_optimizer`` each rank holds the planes' Adam moments for
        its own rows only -- they are gathered here (in place, the same collective as the parameters' all-gather) before the dict is built"""
        if getattr(self, "_sharded", False):
            works = []
            for p in self.renderer.feature_module.plane_tensors():
                st = self.optimizer.state.get(p, {})
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in st:
                        works += Trainer._all_gather_rows(st[key], self.rank, self.world)
            for w in works:
                w.wait()
        self.optimizer.partial_state_reason = None
        try:
            return self.optimizer.state_dict()
        finally:
            if getattr(self, "_sharded", False):
                self.optimizer.partial_state_reason = Trainer._PARTIAL

    _PARTIAL = ("sharded optimizer pass: the planes' Adam moments are valid on the owning rank's rows only -- use "
                "Trainer.optimizer_state_dict(), which gathers them")

    def _planes_adam_early(self, grads) -> None:
        """N == 1 (TN_ADAM_OVERLAP, default on): the planes' gradients are final behind the chain + scatter launch -- their optimizer pass (HBM-bound, no
        LDS, few registers) goes to a stream of its own beside the heads' weight-gradient kernels (VALU / LDS-bound)."""
        pend = getattr(self, "_early_adam", None)
        if pend is None or pend.get("done"):
            return
        if self._side2 is None:
            self._side2 = torch.cuda.Stream(self.device)
        self._side2.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side2):
            self.optimizer.step(plane_reg=pend["plane_reg"], gate=pend["gate"], only="reg")
        pend["done"] = True

    def _planes_ready(self, grads) -> None:
        """Called by the fused render node in the middle of the backward pass (N > 1), as soon as the plane gradients are
        final: their all-reduce starts here and travels while the heads' weight gradients are still being computed."""
        if getattr(self, "_sharded", False):
            for g in grads:
                self._early[g.data_ptr()] = Trainer._reduce_scatter_rows(g, self.rank, self.world)
            return
        views, ptrs = [], []
        for i, g in enumerate(grads):
            flat = self._reduce_view(g, i if self._plane_of else None)
            if flat is not None:
                views.append(flat)
                ptrs.append(g.data_ptr())
        work = Trainer._all_reduce_many(views)
        for k, ptr in enumerate(ptrs):
            self._early[ptr] = work if k == 0 else None          # one handle for the whole group

    @staticmethod
    def _flat_small_grads(params, device):
        """``p.grad`` of every small contiguous parameter as a 256-byte-aligned view into one flat zero buffer; returns
        (buffer, floats in use including the trailing gate slot, ids of the parameters it covers)"""
        small = [p for p in params if p.numel() < (1 << 18) and p.is_contiguous()]
        offs, off = [], 0
        for p in small:
            offs.append(off)
            off = (off + p.numel() + 63) // 64 * 64
        flat = torch.zeros(off + 64, device=device)
        for p, o in zip(small, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        return flat, off + 1, {id(p) for p in small}

    def all_reduce_grads(self, gate: Optional[torch.Tensor] = None) -> None:
        """Sum gradients over ranks with RCCL.  Large plane gradients go as individual in-place all-reduces
        (each drives all xGMI peers; started early by ``_planes_ready`` when the fused path is in use); everything
        small lives in one flat buffer (``_flat``) and travels as ONE in-place all-reduce, together with the step's
        "Empty iteration" gate (in place: > 0 afterwards iff some rank's step reached a parameter -- the step on the union of all
        ranks' rays is empty only when every rank's is)."""
        small, large, handles = [], [], [h for h in self._early.values() if h is not None]
        early, self._early = self._early, {}
        for p in self.renderer.parameters():
            if p.grad is None or id(p) in self._flat_ids:
                continue
            if p.grad.data_ptr() in early:
                continue
            if getattr(self, "_sharded", False) and id(p) in self._plane_of:
                handles.append(Trainer._reduce_scatter_rows(p.grad, self.rank, self.world))      # (not started mid-backward: here)
                continue
            flat = self._reduce_view(p.grad, self._plane_of.get(id(p)))
            if flat is not None:                 # large and dense in memory: reduced in place (live rows only)
                large.append(flat)
            else:
                small.append(p.grad)
        work = Trainer._all_reduce_many(large)
        if work is not None:
            handles.append(work)
        if self._flat is not None:
            slot = self._flat[self._flat_used - 1:self._flat_used]
            if gate is not None:
                torch.sign(gate, out=slot)               # gate = the step's largest weight >= 0 (rewritten every step: no clearing)
            torch.distributed.all_reduce(self._flat[:self._flat_used])
            if gate is not None:
                gate.copy_(slot)
        elif gate is not None:
            flag = (gate > 0).float()
            torch.distributed.all_reduce(flag)
            gate.copy_(flag)
        if small:                                # (small tensors that are not views of the flat buffer: none in the harness)
            bucket = torch.cat([g.reshape(-1) for g in small])
            torch.distributed.all_reduce(bucket)
            off = 0
            for g in small:
                g.copy_(bucket[off:off + g.numel()].view_as(g))
                off += g.numel()
        for h in handles:
            h.wait()

    # ------------------------------------------------------------------ inference (run.py:15-50)
    @torch.no_grad()
    def render_rays(self, rays_o: torch.Tensor, rays_d: torch.Tensor, batch_size: Optional[int] = None) -> torch.Tensor:
        self.renderer.eval()
        # rays are independent: the chunk size does not change a single output bit.  The reference walks an image in chunks of
        # the training batch size (run.py:35-43: 625 calls per 800x800 image), which makes every launch tiny: 139 ms per
        # K-Planes image against 25 ms in 2^16-ray chunks (7 GiB of scratch at S = 1024; scripts/infer_chunks.py)
        bs = batch_size or max(self.cfg.batch_size, 1 << 16)
        out = []
        for k in range(0, rays_o.size(0), bs):
            samples, info = self.ray_provider(rays_o[k:k + bs], rays_d[k:k + bs], training=False)
            out.append(self.renderer(samples, info))
        return torch.cat(out, 0)


# ---------------------------------------------------------------------------------------------------------
# train() counterpart (run.py:97-319): loop, periodic eval, final test render, metrics_*.json, model.pt
# ---------------------------------------------------------------------------------------------------------
@dataclass
class EvalMetrics:
    mse_loss: float = 0.
    psnr: float = 0.
    ssim: float = 0.          # never computed by the reference either (run.py:56-60)


@torch.no_grad()
def infer(trainer: "Trainer", dataset, indices: List[int], folder=None, name: str = "render", batch_size: Optional[int] = None):
    """Render whole images (run.py:15-50); PNGs are written when `folder` is given."""
    out = []
    for i in indices:
        item = dataset[i]
        o, d = item["rays_o"].reshape(-1, 3).to(trainer.device), item["rays_d"].reshape(-1, 3).to(trainer.device)
        img = trainer.render_rays(o, d, batch_size).reshape(*item["rays_o"].shape[:-1], 3)
        out.append(img)
        if folder is not None:
            from PIL import Image
            Image.fromarray((255. * img).clamp(0, 255).to(torch.uint8).cpu().numpy()).save(folder / f"{name}_{i:04d}.png")
    return out


def evaluate(dataset, rendered: List[torch.Tensor], indices: List[int]) -> List[EvalMetrics]:
    """run.py:62-76."""
    res = []
    for i, img in zip(indices, rendered):
        true = dataset[i]["rgbs"].to(img.device)
        res.append(EvalMetrics(mse_loss=torch.nn.functional.mse_loss(true, img).item(), psnr=psnr(true, img).item()))
    return res


def train(cfg: TrainConfig, train_rays, eval_set=None, test_set=None, output=None, eval_every: Optional[int] = None,
          eval_n: int = 1, max_steps: Optional[int] = None, device: Optional[torch.device] = None, log_every: int = 100):
    """The reference's train() on the HIP path.  `train_rays` is a data.RaysDataset on the device."""
    import json
    from dataclasses import asdict
    device = device or train_rays.rays_o.device
    cfg.scene_scale = getattr(train_rays, "scene_scale", cfg.scene_scale)
    tr = Trainer(cfg, train_rays.rays_o, train_rays.rays_d, train_rays.rgbs,
                 None if train_rays.bg_color is None else train_rays.bg_color.to(device), device)
    n_steps = tr.steps if max_steps is None else min(tr.steps, max_steps)
    eval_metrics, eval_step = [], 0
    # {loss, occupancy} of EVERY step like the reference (run.py:262-266), kept on the device and read back in one copy
    log = torch.zeros((n_steps + 1, 2), dtype=torch.float64, device=device)
    for step in range(n_steps + 1):                       # the reference runs steps+1 iterations (run.py:290)
        tr.step()
        log[step, 0] = tr.loss_device()[0]
        log[step, 1] = tr.occupancy_grid._stats_device()[1] / tr.occupancy_grid.grid.numel()
        if step % log_every == 0 or step == n_steps:
            lv, ov = log[step].tolist()
            print(f"step {step}/{n_steps} loss {lv:.5f} occupancy {ov:.3f} samples {int(tr.last['n_samples'])}")
        if eval_every and eval_set is not None and step % eval_every == 0 and step > 0:
            idx = list(range(eval_step, min(eval_step + eval_n, len(eval_set))))
            eval_metrics.extend(asdict(m) for m in evaluate(eval_set, infer(tr, eval_set, idx, output, f"test_{step}"), idx))
            eval_step += eval_n
    test_metrics = None
    if test_set is not None:
        idx = list(range(len(test_set)))
        rendered = infer(tr, test_set, idx, output, "test_full")
        if test_set.rgbs:
            test_metrics = [asdict(m) for m in evaluate(test_set, rendered, idx)]
    train_metrics = [{"loss": lv, "occupancy": ov} for lv, ov in log.tolist()]
    if output is not None:
        torch.save(tr.renderer.state_dict(), output / "model.pt")            # run.py:308
        json.dump(train_metrics, open(output / "metrics_train.json", "w"))
        if eval_metrics:
            json.dump(eval_metrics, open(output / "metrics_eval.json", "w"))
        if test_metrics:
            json.dump(test_metrics, open(output / "metrics_test.json", "w"))
    return tr, train_metrics, eval_metrics, test_metrics
