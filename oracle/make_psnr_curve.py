#!/usr/bin/env python3
"""PSNR@step of the reference's recipe on CPU (build container only) -> tests/golden/G17_psnr_curve.json.

TEST INFRASTRUCTURE.  Runs ``oracle/torch_port.reference_training`` -- the CPU port of the reference's ``train()`` (run.py:97-319:
shuffled loader stream, sampling jitter, jittered occupancy refreshes every 16 * 4096 / B steps, Adam + MultiStepLR, the scaled and
never unscaled loss) -- on the synthetic scene for N_STEPS steps and several seeds, and records the held-out PSNR (run.py:53-54,
``infer`` semantics: training=False sampling, run.py:15-50) at EVAL_AT.  ``tests/test_hip_psnr.py`` (-m gpu) runs the HIP
``Trainer`` (device RNG, refreshes on) on the same scene from the same initial parameters and holds the seed-mean curves together.

    python oracle/make_psnr_curve.py [--seeds 0 1 2] [--steps 300] [--out tests/golden/G17_psnr_curve.json]
    python oracle/make_psnr_curve.py --replay --seeds 0 1 --out tests/golden/G18_psnr_replay.json
    python oracle/make_psnr_curve.py --replay --seeds 0 --method vanilla --lr 1e-3 --steps 200     (G19; cobafa: G20)
    python oracle/make_psnr_curve.py --bench                                                        (G21: the bench's own configuration)

``--bench`` (round 5): the replay run on the configuration BASELINE.json's metric is quoted on -- bench.py's 20 synthetic 800 x 800
training cameras (12.8 M rays), B = 1024, S = 1024, its occupancy ball as the initial grid, seed 0, 70 steps -- with the held-out PSNR of
bench.py's 800 x 800 test camera (``rays.synthetic_scene(n_views=1, seed=10007)``) at the step counts the bench ends on (65 with the
driver's ``--warmup 5 --steps 20``, 70 with the defaults).  ~45 min on 8 cores -> tests/golden/G21_psnr_bench.json;
``bench.py`` reports ``psnr_at_step.reference`` / ``delta_db`` from it and tests/test_hip_psnr.py holds the HIP trainer against it.

``--replay``: every random choice (ray order, sampling jitter, refresh jitter) comes from the streams the HIP harness defines for
``TrainConfig(seed, host_shuffle=True)`` (restated in ``reference_training(replay=...)``), so the GPU run walks the same rays with the
same jitter and the two PSNR curves can be held together directly -- no seed statistics (G18); without it the port draws from its
own numpy generator, as two machines running the reference would (G17: compared through seed means).

The scene is ``tinynerf_amd.rays.synthetic_scene`` (an input generator shared with the GPU test, not a product path).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tinynerf_oracle as orc          # noqa: E402
from oracle import torch_port as tp                 # noqa: E402

CONFIG = dict(n_views=21, res=100, batch_size=1024, n_samples=128, occupancy_res=128, method="kplanes", scene_seed=0)
EVAL_AT = (0, 50, 100, 150, 200, 250, 300)
EVAL_AT_REPLAY = (0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 125, 150, 200, 250, 300)
# the bench's own workload (bench.py main(): TrainConfig(kplanes, aabb, 1024, 1024, seed 0), 20 views, occupancy ball)
BENCH_CONFIG = dict(n_views=20, res=800, batch_size=1024, n_samples=1024, occupancy_res=128, method="kplanes", scene_seed=0,
                    heldout_seed=10_007, grid0="ball(0.5) / decay^20")
BENCH_EVAL_AT = (0, 30, 65, 70)


def bench_grid0(res: int = 128) -> np.ndarray:
    """bench.py's initial occupancy grid: 1 inside the centred ball of radius 0.5 (normalised coordinates), decay^20 elsewhere --
    built with torch's CPU kernels on both sides (bench.py uploads this very array)"""
    decay = 0.01 ** (1 / 16)
    lin = torch.linspace(-1, 1, res)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, decay ** 20).to(torch.float32).numpy()


def bench_main(args):
    """G21: see the module docstring"""
    from tinynerf_amd import rays
    c = BENCH_CONFIG
    torch.set_num_threads(args.threads)
    o, d, rgbs, _, _ = rays.synthetic_scene(n_views=c["n_views"], res=c["res"], seed=c["scene_seed"], device="cpu")
    ho, hd, hrgb, _, _ = rays.synthetic_scene(n_views=1, res=c["res"], seed=c["heldout_seed"], device="cpu")
    o, d, rgbs, ho, hd = o.numpy(), d.numpy(), rgbs.numpy(), ho.numpy(), hd.numpy()
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    bg = torch.ones(3)
    steps = args.steps if args.steps != 300 else 70
    eval_at = [e for e in BENCH_EVAL_AT if e <= steps]
    out = args.out or os.path.join(ROOT, "tests", "golden", "G21_psnr_bench.json")
    curve, t0 = {}, time.perf_counter()

    def eval_fn(step, sd, grid, thr):
        sdd = {k: v.detach() for k, v in sd.items()}
        parts, n_s = [], 0
        for k in range(0, ho.shape[0], 8192):               # (the numpy sampler materialises [R, S, 3] arrays: chunks of 8192 rays)
            packed, info = orc.ray_provider(ho[k:k + 8192], hd[k:k + 8192], marcher="aabb", contraction="aabb", grid=grid, threshold=thr,
                                            n_samples=c["n_samples"], near=0.1, aabb=aabb)
            n_s += packed.shape[0]
            with torch.no_grad():
                parts.append(tp.render(sdd, torch.from_numpy(packed), torch.from_numpy(info), bg))
        img = torch.cat(parts, 0)
        curve[step] = float(-10.0 * torch.log10(torch.mean((img - hrgb) ** 2)))
        print(f"step {step}: held-out psnr {curve[step]:.4f} dB, {n_s} samples ({time.perf_counter() - t0:.0f} s)", flush=True)
        json.dump(curve, open(out + ".partial", "w"))

    def on_step(step, sd, packed, info, target):
        print(f"step {step}: {packed.shape[0]} samples / {info.shape[0]} rays ({time.perf_counter() - t0:.0f} s)", flush=True)
    sd0 = initial_state(0, c["method"], batch_size=c["batch_size"], n_samples=c["n_samples"])
    losses, _, counts = tp.reference_training(sd0, o, d, rgbs, method=c["method"], batch_size=c["batch_size"], n_samples=c["n_samples"],
                                              n_steps=steps, occupancy_res=c["occupancy_res"], grid0=bench_grid0(c["occupancy_res"]),
                                              eval_at=eval_at, eval_fn=eval_fn, on_step=on_step, replay={"seed": 0, "rank": 0})
    json.dump({"config": c, "eval_at": eval_at, "steps": steps, "torch": torch.__version__, "replay": True,
               "runs": [{"seed": 0, "psnr": {str(k): v for k, v in sorted(curve.items())}, "loss": losses,
                         "samples_per_step": [cn[0] for cn in counts], "rays_per_step": [cn[1] for cn in counts]}],
               "made_by": "oracle/make_psnr_curve.py --bench (CPU port of the reference's train() on bench.py's configuration, replay streams)"},
              open(out, "w"), indent=1)
    if os.path.exists(out + ".partial"):
        os.remove(out + ".partial")


def scene():
    from tinynerf_amd import rays
    o, d, rgbs, _, _ = rays.synthetic_scene(n_views=CONFIG["n_views"], res=CONFIG["res"], seed=CONFIG["scene_seed"], device="cpu")
    per = CONFIG["res"] ** 2
    n_train = (CONFIG["n_views"] - 1) * per
    return (o[:n_train], d[:n_train], rgbs[:n_train]), (o[n_train:], d[n_train:], rgbs[n_train:])


def initial_state(seed: int, method: str = None, batch_size: int = None, n_samples: int = None):
    """the parameters a run with ``seed`` starts from: the reference's constructors under ``torch.manual_seed(seed)`` (run.py:130-152),
    restated in ``oracle/torch_port.initial_state`` and pinned bit for bit to the reference's own ``train()`` by golden G22
    (tests/test_oracle_train_trace.py).  The HIP harness seeds its constructors the same way (``run.Trainer``), so both sides start
    from identical parameters without the checker importing the product's constructor (rounds 4 - 5 did)."""
    return {k: v.detach().clone().contiguous() for k, v in tp.initial_state(method or CONFIG["method"], seed).items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--replay", action="store_true")
    ap.add_argument("--method", default=CONFIG["method"], choices=["kplanes", "vanilla", "cobafa"])
    ap.add_argument("--lr", type=float, default=1e-2, help="run.py:110 has 1e-2; the two deep stacks fall into the all-masked branch "
                    "with it on this scene within a few steps (in the port as on the GPU): their curves use 1e-3")
    ap.add_argument("--out", default=None)
    ap.add_argument("--bench", action="store_true", help="G21: the replay run on bench.py's own configuration (800 x 800, B = S = 1024)")
    ap.add_argument("--threads", type=int, default=torch.get_num_threads())
    args = ap.parse_args()
    if args.bench:
        return bench_main(args)
    cfg = dict(CONFIG, method=args.method)
    if args.lr != 1e-2:
        cfg["lr"] = args.lr
    default = {"kplanes": "G18_psnr_replay.json", "vanilla": "G19_psnr_replay_vanilla.json", "cobafa": "G20_psnr_replay_cobafa.json"}[args.method]
    args.out = args.out or os.path.join(ROOT, "tests", "golden", default if args.replay else "G17_psnr_curve.json")
    vf = 10 if args.method == "vanilla" else 0
    cf = torch.linspace(2., 8., 6).tolist() if args.method == "cobafa" else None      # run.py:144 (Dropout(0.01) off on both sides)
    (o, d, rgbs), (ho, hd, hrgb) = scene()
    o, d, rgbs = o.numpy(), d.numpy(), rgbs.numpy()
    ho, hd = ho.numpy(), hd.numpy()
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    bg = torch.ones(3)
    eval_at = [e for e in (EVAL_AT_REPLAY if args.replay else EVAL_AT) if e <= args.steps]
    runs = []
    for seed in args.seeds:
        curve = {}

        def eval_fn(step, sd, grid, thr, curve=curve):
            packed, info = orc.ray_provider(ho, hd, marcher="aabb", contraction="aabb", grid=grid, threshold=thr,
                                            n_samples=CONFIG["n_samples"], near=0.1, aabb=aabb)        # training=False: no jitter
            with torch.no_grad():
                img = tp.render({k: v.detach() for k, v in sd.items()}, torch.from_numpy(packed), torch.from_numpy(info), bg,
                                vanilla_freqs=vf, cobafa_freqs=cf)
            curve[step] = float(-10.0 * torch.log10(torch.mean((img - hrgb) ** 2)))
            print(f"seed {seed} step {step}: held-out psnr {curve[step]:.3f} dB ({time.perf_counter() - t0:.0f} s)", flush=True)
        t0 = time.perf_counter()
        losses, _, counts = tp.reference_training(initial_state(seed, args.method), o, d, rgbs, method=args.method, batch_size=CONFIG["batch_size"],
                                                  n_samples=CONFIG["n_samples"], n_steps=args.steps, occupancy_res=CONFIG["occupancy_res"],
                                                  eval_at=eval_at, eval_fn=eval_fn, lr=args.lr, cobafa_freqs=cf,
                                                  **({"replay": {"seed": seed, "rank": 0}} if args.replay else {"stochastic_seed": 1000 + seed}))
        runs.append({"seed": seed, "psnr": {str(k): v for k, v in sorted(curve.items())}, "loss": losses,
                     "samples_per_step": [c[0] for c in counts], "rays_per_step": [c[1] for c in counts]})
        json.dump({"config": cfg, "eval_at": eval_at, "steps": args.steps, "torch": torch.__version__, "runs": runs, "replay": bool(args.replay),
                   "made_by": "oracle/make_psnr_curve.py (CPU port of the reference's train(), stochastic mode)"},
                  open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
