#!/usr/bin/env python3
"""Trace of the reference's OWN train() loop (src/run.py:97-319) -> tests/golden/G22_reference_train_*.npz.

TEST INFRASTRUCTURE ONLY -- runs in the build container (where /root/reference exists), never on the GPU box.

What runs is the reference's code, imported from a scratch copy: ``src.run.train(cfg)`` with its own model constructors, its
``DataLoader(shuffle=True)``, dynamic-batch loop, ``OccupancyGrid.update``, ``NerfRenderer``, ``torch.optim.Adam`` +
``ChainedScheduler([MultiStepLR])`` and the scaled-and-never-unscaled loss.  Five things are stubbed, each because the build
container cannot run the original, and none of them is arithmetic of the path:

* ``torch.utils.cpp_extension.load`` (core.py:7) -> ``oracle/weights_ref.c`` (as ``oracle/make_goldens.py`` does: cuda.cu needs nvcc);
* ``torch.cuda.amp.GradScaler`` (run.py:201) -> a stand-in whose ``scale(loss)`` multiplies by the initial scale.  That is what the
  real class does on a CUDA device; on a CPU-only host it disables itself (scale() returns the loss unchanged) and the quirk of
  run.py:259-260 -- the loss is scaled and ``optimizer.step()`` is called on the raw optimizer, never ``scaler.step`` -- would go
  untested;
* the three random streams -- ``torch.randperm`` of the loader's ``RandomSampler``, ``torch.rand_like`` of the sampling jitter
  (core.py:173) and of the refresh jitter (core.py:136) -- are replaced by streams that can be regenerated ANYWHERE from one
  integer: successive ``torch.randperm(M, generator=Generator().manual_seed(s * 1000003 + 1), dtype=int32)`` permutations and the
  counter RNG ``oracle.tinynerf_oracle.uniform01`` with the seeds of ``tinynerf_amd.run.jitter_seed / refresh_seed`` (restated
  below).  A 40-step trace would otherwise have to carry every batch (150 MB); this way it carries seeds, and the checker
  (``oracle/torch_port.reference_training(replay=..., loader="dataloader")``) and the HIP harness re-walk the same rays with the
  same jitter;
* ``DataLoader``'s ``num_workers=8, pin_memory=True`` -> ``0, False`` (same batches: the sampler runs in the parent process);
* ``tqdm`` -> a recorder that captures ``set_postfix(loss, occupancy, rendered_samples)`` and stops the loop after K steps.

Recorded per step: loss (run.py:264, the UNSCALED MSE + TV), occupancy fraction, packed samples, rays, loader batches, learning rate
after ``scheduler.step()``; per refresh: the whole grid as decay counts (a cell is 1 or fl(decay * ...), core.py:140-144) and its
mean; sha256 of every initial tensor (the constructors under ``torch.manual_seed(seed)``: what ``oracle/torch_port.initial_state``
must reproduce bit for bit); the final ``state_dict`` -- every small tensor in full, planes / wide layers as a fixed strided subsample
plus fp64 sums.

    python oracle/make_train_trace.py                       # G22_reference_train_{kplanes,vanilla,kplanes_lr}.npz
    python oracle/make_train_trace.py --only kplanes
"""
import argparse
import hashlib
import os
import shutil
import sys
import tempfile
import time
import types

import numpy as np

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

# name -> (method, batch_size, n_samples, steps recorded, seed).  B = 4096: bs_ratio = 1, i.e. the recipe's own 2048 steps /
# refresh every 16 (run.py:100-103).  "kplanes_lr": B = 65536 -> 128 steps, refresh EVERY step, milestones 64 / 96 / 106 / 115
# (run.py:188-199): 70 recorded steps cross the first one, so the scheduler's arithmetic is in the trace.
TRACES = {
    "kplanes": ("kplanes", 4096, 32, 21, 1),
    "vanilla": ("vanilla", 4096, 32, 21, 2),
    "kplanes_lr": ("kplanes", 65536, 4, 70, 3),
}
SUBSAMPLE_OVER = 1 << 16          # tensors larger than this are recorded as a strided subsample + sums


def jitter_seed(seed: int, batch_no: int, rank: int = 0) -> int:
    """restatement of tinynerf_amd.run.jitter_seed (the oracle may not import the product)"""
    h = int.from_bytes(hashlib.blake2b(b"tinynerf-jitter:%d:%d" % (seed, batch_no), digest_size=8).digest(), "little")
    return (h & (2 ** 62 - 1)) * 2 + 1 + rank


def refresh_seed(seed: int, train_step: int) -> int:
    """restatement of tinynerf_amd.run.refresh_seed"""
    return (seed * 7919 + 104729 * (train_step + 1)) % (2 ** 62)


def subsample_index(numel: int) -> np.ndarray:
    """the fixed positions at which a large tensor is recorded (flat index into the CONTIGUOUS state_dict tensor)"""
    stride = max(1, numel // 8192)
    return np.arange(0, numel, stride, dtype=np.int64)[:8192]


class StopTrace(Exception):
    pass


def run_trace(name, ref_dir, out_dir):
    import torch
    import torch.utils.cpp_extension as ce
    import tinynerf_oracle as orc
    method, B, S, K, seed = TRACES[name]
    scratch = tempfile.mkdtemp(prefix="tinynerf_ref_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(ref_dir, dst)

    def fwd(s, d, info, thr):
        return torch.from_numpy(orc.weights_fwd(s.detach().numpy(), d.detach().numpy(), info.numpy(), float(thr)))

    def bwd(s, d, info, w, g):
        return torch.from_numpy(orc.weights_bwd(s.detach().numpy(), d.detach().numpy(), info.numpy(), w.detach().numpy(), g.detach().numpy()))
    ce.load = lambda *a, **k: types.SimpleNamespace(compute_weights_fwd=fwd, compute_weights_bwd=bwd)

    class ScalerStandIn:                  # torch.cuda.amp.GradScaler(2 ** 10) as it behaves on a CUDA device, used as run.py uses it
        def __init__(self, init_scale=2.0 ** 16, *a, **k):
            self._s = float(init_scale)

        def scale(self, loss):
            return loss * self._s
    torch.cuda.amp.GradScaler = ScalerStandIn
    for m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[m]
    sys.path.insert(0, dst)
    import warnings
    warnings.filterwarnings("ignore")
    cwd = os.getcwd()
    os.chdir(dst)                          # (core.py:7 names 'src/cuda.cu' relative to the working directory; load is stubbed anyway)
    import src.core as core
    import src.run as run
    import src.data as data
    from pathlib import Path

    rec = dict(loss=[], occupancy=[], n_samples=[], n_rays=[], k=[], lr=[], grids=[], grid_means=[], grid_steps=[])
    state = dict(step=0, b=0, mode=None, slice=0, opt=None, renderer=None, rays_in_step=0)
    M_holder = {}

    # ---- random streams -------------------------------------------------------------------------------------------------
    host_gen = torch.Generator().manual_seed(seed * 1000003 + 1)
    real_randperm, real_rand_like = torch.randperm, torch.rand_like

    def randperm_stub(n, *a, **k):
        # RandomSampler.__iter__ calls randperm once per epoch and once more for an empty tail slice: both get the epoch's permutation,
        # which LoaderRec.__iter__ (below) drew when the reference's loop restarted its iterator (run.py:221-225)
        if state["mode"] != "train" or n != M_holder.get("M"):
            return real_randperm(n, *a, **k)
        return state["perm"]

    def rand_like_stub(x, *a, **k):
        if state["mode"] == "sample":      # RayProvider.__call__, core.py:173: t_values [R, S]
            R, S_ = x.shape
            r0 = state["b"] * B
            ctr = (np.arange(r0, r0 + R, dtype=np.uint64)[:, None] * np.uint64(S_) + np.arange(S_, dtype=np.uint64)[None, :])
            return torch.from_numpy(orc.uniform01(jitter_seed(seed, state["step"]), ctr))
        if state["mode"] == "refresh":     # OccupancyGrid.update, core.py:136: coords[i] [H, W, 3]
            per = x.numel()
            i = state["slice"]
            state["slice"] += 1
            return torch.from_numpy(orc.uniform01(refresh_seed(seed, state["step"]), np.uint64(i * per) + np.arange(per, dtype=np.uint64)).reshape(tuple(x.shape)))
        return real_rand_like(x, *a, **k)
    torch.randperm, torch.rand_like = randperm_stub, rand_like_stub

    # ---- recorders around the reference's own objects ---------------------------------------------------------------------
    provider_call = core.RayProvider.__call__

    def provider_rec(self, rays_o, rays_d, training):
        prev, state["mode"] = state["mode"], "sample" if training else state["mode"]
        try:
            out = provider_call(self, rays_o, rays_d, training)
        finally:
            state["mode"] = prev
        state["b"] += 1
        state["rays_in_step"] += rays_o.size(0)
        return out
    core.RayProvider.__call__ = provider_rec
    grid_update = core.OccupancyGrid.update
    decay = 0.01 ** (1 / 16)

    def update_rec(self, sigma_fn):
        prev, state["mode"], state["slice"] = state["mode"], "refresh", 0
        try:
            grid_update(self, sigma_fn)
        finally:
            state["mode"] = prev
        g = self.grid.numpy()
        # a cell is 1 or decay * (decay * ...) in fp32 (core.py:140-144): record the number of decays, check the reconstruction
        ladder = [np.float32(1.0)]
        for _ in range(255):
            ladder.append(np.float32(np.float32(self.decay) * ladder[-1]))
        ladder = np.array(ladder, np.float32)
        kk = np.searchsorted(-ladder, -g.ravel()).astype(np.uint8).reshape(g.shape)
        assert np.array_equal(ladder[kk], g), "occupancy cell outside the decay ladder"
        rec["grids"].append(kk)
        rec["grid_means"].append(float(self.mean))
        rec["grid_steps"].append(state["step"])
    core.OccupancyGrid.update = update_rec
    fwd_render = core.NerfRenderer.forward

    def render_rec(self, packed, info, *a, **k):
        state["renderer"] = self
        if self.training:
            rec["n_samples"].append(int(packed.size(0)))
            rec["n_rays"].append(int(info.size(0)))
            rec["k"].append(state["b"])
            assert state["rays_in_step"] == info.size(0)
        return fwd_render(self, packed, info, *a, **k)
    core.NerfRenderer.forward = render_rec

    real_adam = torch.optim.Adam

    class AdamRec(real_adam):
        def __init__(self, params, **kw):
            params = list(params)
            # the initial parameters, before anything has touched them (run.py:186 is the first thing behind the constructors)
            rec["init_names"] = None
            state["init_params"] = [p.detach().clone() for p in params]
            super().__init__(params, **kw)
            state["opt"] = self
            rec["adam_kw"] = {k: float(v) for k, v in kw.items()}
    torch.optim.Adam = AdamRec            # (run.py:186 looks the class up through the module attribute at call time)

    class TqdmRec:
        def __init__(self, it=None, total=None, **k):
            self.it, rec["total_steps"] = it, total

        def __iter__(self):
            return iter(self.it)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def set_postfix(self, loss=None, occupancy=None, rendered_samples=None, **k):
            rec["loss"].append(float(loss))
            rec["occupancy"].append(float(occupancy))
            rec["lr"].append(float(state["opt"].param_groups[0]["lr"]))

        def update(self, n=1):
            state["step"] += n
            state["b"], state["rays_in_step"] = 0, 0
            if state["step"] >= K:
                raise StopTrace()
    run.tqdm = TqdmRec
    real_loader = run.DataLoader

    class LoaderRec(real_loader):
        def __iter__(self):
            state["perm"] = real_randperm(M_holder["M"], generator=host_gen, dtype=torch.int32).to(torch.int64)
            rec["epochs"] = rec.get("epochs", 0) + 1
            return super().__iter__()
    run.DataLoader = lambda **kw: LoaderRec(**dict(kw, num_workers=0, pin_memory=False))

    # ---- the reference's configuration on its own fixture -------------------------------------------------------------------
    train_rays = data.RaysDataset(data.parse_nerf_synthetic(Path("tests/dummy/hotdog"), "train"))
    M_holder["M"] = len(train_rays)
    # the ray table the loop draws from, as the reference's RaysDataset holds it (data.py:109-118): directions in full, origins per
    # image (constant inside one), colours as the 8-bit values they were divided from (data.py:152-153: exact to reconstruct)
    rays_path = os.path.join(out_dir, "G22_rays_hotdog.npz")
    per = train_rays.rays_o.size(0) // 2
    o_img = train_rays.rays_o[::per].numpy().copy()
    assert torch.equal(train_rays.rays_o, torch.from_numpy(o_img).repeat_interleave(per, 0))
    rgb8 = torch.round(train_rays.rgbs * 255.).to(torch.uint8)
    assert torch.equal(rgb8.float() / 255., train_rays.rgbs)
    np.savez_compressed(rays_path, rays_o_per_image=o_img, rays_per_image=per, rays_d=train_rays.rays_d.numpy(), rgbs_u8=rgb8.numpy(),
                        bg_color=train_rays.bg_color.numpy(), scene_scale=float(train_rays.scene_scale))
    out_tmp = Path(tempfile.mkdtemp(prefix="tinynerf_trace_out_"))
    cfg = run.TrainConfig(method=method, train_rays=train_rays, eval_set=None, eval_every=None, eval_n=None, test_set=None,
                          scene_type="aabb", output=out_tmp, batch_size=B, n_samples=S)
    torch.manual_seed(seed)                # train.py:68-72 (SEED): the constructors draw from the global generator
    t0 = time.perf_counter()
    state["mode"] = "train"
    try:
        run.train(cfg)
        raise AssertionError("the reference's loop ended before the trace did")
    except StopTrace:
        pass
    finally:
        state["mode"] = None
        torch.randperm, torch.rand_like, torch.optim.Adam = real_randperm, real_rand_like, real_adam
        os.chdir(cwd)
        sys.path.remove(dst)
    print(f"{name}: {K} steps of src.run.train() in {time.perf_counter() - t0:.0f} s; losses {rec['loss'][:3]} ... {rec['loss'][-2:]}")

    renderer = state["renderer"]
    names = [n for n, _ in renderer.named_parameters()]
    out = dict(method=method, batch_size=B, n_samples=S, steps=K, seed=seed, total_steps=rec["total_steps"], n_rays_table=M_holder["M"],
               epochs_started=rec["epochs"],
               loss=np.array(rec["loss"], np.float64), occupancy=np.array(rec["occupancy"], np.float64),
               n_samples_per_step=np.array(rec["n_samples"], np.int64), n_rays_per_step=np.array(rec["n_rays"], np.int64),
               loader_batches_per_step=np.array(rec["k"], np.int64), lr_after_step=np.array(rec["lr"], np.float64),
               grid_steps=np.array(rec["grid_steps"], np.int64), grid_means=np.array(rec["grid_means"], np.float64),
               grid_decays=np.stack(rec["grids"]), param_names=np.array(names),
               adam_lr=rec["adam_kw"]["lr"], adam_eps=rec["adam_kw"]["eps"], adam_weight_decay=rec["adam_kw"]["weight_decay"],
               torch_version=torch.__version__)
    assert len(names) == len(state["init_params"])
    for n, p0, p in zip(names, state["init_params"], renderer.parameters()):
        a0 = np.ascontiguousarray(p0.numpy())
        a = np.ascontiguousarray(p.detach().numpy())
        out["init_sha256/" + n] = hashlib.sha256(a0.tobytes()).hexdigest()
        out["shape/" + n] = np.array(a.shape, np.int64)
        if a.size > SUBSAMPLE_OVER:
            idx = subsample_index(a.size)
            out["final_sub/" + n] = a.ravel()[idx]
            out["init_sub/" + n] = a0.ravel()[idx]
            out["final_sum/" + n] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum(),
                                              (a.astype(np.float64) ** 2).sum()])
        else:
            out["final/" + n] = a
    path = os.path.join(out_dir, f"G22_reference_train_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")
    shutil.rmtree(scratch, ignore_errors=True)
    shutil.rmtree(out_tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", nargs="*", default=None, choices=list(TRACES))
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    import torch
    torch.set_num_threads(args.threads)
    os.makedirs(os.path.abspath(args.out), exist_ok=True)
    for name in (args.only or list(TRACES)):
        run_trace(name, args.ref, os.path.abspath(args.out))


if __name__ == "__main__":
    main()
