#!/usr/bin/env python3
"""Headline benchmark: ray-samples/sec of the K-Planes training step on synthetic Lego-shaped input.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
K-Planes field (3 scales x 3 planes of 32 channels, 128^2/256^2/512^2) + Vanilla sigma / colour
decoders, aabb +-1.5, S = 1024 candidates per ray, B = 1024 rays per loader batch, dynamic batches of
~B*S = 2^20 packed samples per step, 128^3 occupancy grid occupied inside a centred ball of radius 0.5
(normalised), 800x800 pinhole cameras on the radius-4.0311 sphere, synthetic colours.  All inputs are
resident in HBM before the timed region.

A "step" is one full pass of the hot path as train() drives it (reference run.py:215-261): dynamic
batch assembly by the sampler, render forward, MSE (+TV) loss, backward through every kernel, Adam
step, LR scheduler step (+ gradient all-reduce when N > 1, + the occupancy refresh whenever the
reference's schedule puts one inside the window).  value = packed samples processed by all ranks /
wall time of exactly K steps (barrier + synchronize on both sides, max over ranks).

The line also carries `roofline` for the dominant kernel (HIP-event timed on its own stream) and
`cpu_baseline` (the CPU port of the same step, oracle/torch_port.py, on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

torch = None     # imported in main(), AFTER the launcher branch: the parent that starts the ranks never loads it

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0   # fp32-equivalent FLOP/s of the exact 3-way split: six bf16 products per fp32 product
PEAK_F16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0    # ... of the two-term fp16 split: three fp16 products (v_mfma_f32_32x32x16_f16 runs at the bf16 rate)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
PEAK_ATOMIC_GLANES = 325.0         # scripts/microbench/atomic_scaling.hip: full-line global_atomic_add_f32 from >= 4096 waves on random lines, G lane-atomics/s
                                   # (atomic_patterns.hip: 272 on random lines with its heavier index stream, 160 on runs of neighbouring lines)
# counter passes cannot be collected live (rocprofv3 wraps the process): the committed summaries of the same command
PMC_PROFILES = ["profiles/round6_pmc_traffic.json", "profiles/round5_pmc_traffic.json", "profiles/round4_pmc_traffic.json", "profiles/round3_pmc_traffic.json", "profiles/round2_pmc_traffic.json"]   # scripts/pmc.sh + scripts/pmc_to_json.py
MFMA_PROFILES = ["profiles/round6_mfma_busy.json", "profiles/round5_mfma_busy.json", "profiles/round4_mfma_busy.json", "profiles/round3_mfma_busy.json", "profiles/round2_mfma_busy.json"]      # scripts/pmc_mfma.sh

# algorithmic work per unit of the kernels that can dominate
FLOP_SIGMA_FWD = 2 * (96 * 64 + 64)                                   # 12 416   
FLOP_RGB_FWD = 2 * (147 * 64 + 3 * 64 * 64 + 64 * 3)                  # 43 776   
FLOP_HEADS = FLOP_RGB_FWD + FLOP_SIGMA_FWD                            # 56 192
# the data gradient of the colour head's first layer stops at the 96 feature columns: its 51 per-ray columns [PE(d), d]
# are inputs, not parameters (models.py:87), so W_0^T G_0 is 96 x 64, not 147 x 64 -> 2 * 51 * 64 = 6 528 FLOP less
FLOP_HEADS_DGRAD = FLOP_HEADS - 2 * 51 * 64                           # 49 664 (= 383 MFMAs per 32-sample tile: the MFMA_BUSY count)
FLOP_STEP = FLOP_HEADS + FLOP_HEADS_DGRAD + FLOP_HEADS                # forward + data gradient + weight gradient = 162 048
KERNEL_MODEL = {
    # timed tag              (bound, unit work per row, what a row is, kernels of the launch)
    "tn_mlp_bwd:rgb": ("mfma", 2 * FLOP_RGB_FWD, "active sample", "chain + weight-gradient kernels of the colour head"),
    "tn_mlp_bwd:sigma": ("mfma", 2 * FLOP_SIGMA_FWD, "sample", "chain + weight-gradient kernels of the sigma head"),
    "tn_mlp_fwd:rgb": ("mfma", FLOP_RGB_FWD, "active sample", "mlp_fwd_kernel"),
    "tn_mlp_fwd:sigma": ("mfma", FLOP_SIGMA_FWD, "sample", "mlp_fwd_kernel"),
    "tn_mlp_bwd_pair": ("mfma", FLOP_HEADS_DGRAD + FLOP_HEADS, "sample", "mlp_chain_kernel + mlp_wgrad4_kernel + mlp_wgrad_kernel"),
    "tn_mlp_fwd_stash_pair": ("mfma", FLOP_HEADS, "sample", "mlp_fwd_kernel (both heads)"),
    # gather + both heads' training forward in one kernel: MFMA-bound (36 texel lines per sample ride under the MFMAs)
    "tn_kplanes_mlp_fwd_pair": ("mfma", FLOP_HEADS, "sample", "mlp_fwd_kernel<..., KP> (gather + both heads, one kernel)"),
    # data-gradient chain of both heads + the scatter into the nine plane gradients in one kernel: bound by the rate of
    # memory-side fp32 atomics (the lane-atomic count per launch comes from the PMC pass: WRITE_SIZE / 4 B)
    "tn_kplanes_mlp_bwd_pair:chain": ("atomic", FLOP_HEADS_DGRAD, "sample", "mlp_chain_kernel<..., KP> (both chains + plane scatter, one kernel)"),
    "tn_kplanes_mlp_bwd_pair:wgrad": ("mfma", FLOP_HEADS, "sample", "wgrad_first_kernel + wgrad_rc_kernel (TN_MLP_LEAN: activations rebuilt) | mlp_wgrad4_kernel + mlp_wgrad_kernel (stash form)"),
    "tn_kplanes_fwd": ("hbm", 12 + 4608 + 384, "sample", "kplanes_fwd_kernel"),
    "tn_kplanes_bwd": ("atomic", 0, "sample", "kplanes_bwd_kernel"),
    "tn_adam_reg_multi": ("hbm", 32, "plane element", "adam_reg_multi_kernel (p, g, m, v in; p, g, m, v out)"),
}


# kernel-name prefixes (scripts/pmc_mfma.sh output) behind each timed tag, for the counter-derived matrix-pipe fraction
MFMA_KERNELS = {
    "tn_kplanes_mlp_fwd_pair": ["mlp_fwd_kernel<64, true, 12, true, true, true, true", "mlp_fwd_kernel<64, true, 8, true, true, true, true"],
    "tn_kplanes_mlp_bwd_pair:chain": ["mlp_chain_kernel<64, 4, 8, true, false, true, true"],
    "tn_kplanes_mlp_bwd_pair:wgrad": ["mlp_wgrad4_kernel<4", "mlp_wgrad_kernel<64, 1", "wgrad_first_kernel", "wgrad_rc_kernel"],
}
# Which matrix instructions a timed launch issues, and the dense peak of THAT class in fp32-equivalent TFLOP/s (a fraction of the fp32 MFMA
# peak says nothing about a launch that runs on the fp16 / bf16 cores).  f16x2: 3 fp16 MFMAs per fp32 product block; bf16x3: 6 bf16 MFMAs.
MFMA_CLASS = {
    "f16x2": ("v_mfma_f32_32x32x16_f16, two-term fp16 splits", PEAK_F16X2_TFLOPS),
    "bf16x3": ("v_mfma_f32_32x32x16_bf16, three-term bf16 splits", PEAK_BF16X3_TFLOPS),
    "fp32": ("v_mfma_f32_32x32x2_f32", PEAK_FP32_MFMA_TFLOPS),
}


def launch_class(tag: str, mode: str, lean: bool):
    """(instruction class, what actually limits the launch) of a timed tag under matrix mode `mode`"""
    if tag == "tn_kplanes_mlp_fwd_pair":
        return ("f16x2" if mode == "f16x2" else "fp32",
                "gather round trips + VALU (lean: 0.45 KB written per sample)" if lean else "HBM writes of the activation workspace (2.8 KB per sample)")
    if tag == "tn_kplanes_mlp_bwd_pair:chain":
        return ("fp32", "memory-side fp32 atomics of the plane scatter (co_bound)")
    if tag == "tn_kplanes_mlp_bwd_pair:wgrad":
        return (("f16x2 (rebuilt forward, both orientations) + bf16x3 (G x H) + fp32 (first layers)", "VALU issue + matrix pipe of one wave per SIMD (wgrad_rc), fp32 matrix pipe (wgrad_first)") if lean
                else ("fp32", "HBM reads of the activation workspace (3.7 KB per sample)"))
    return ("fp32", None)


def _matmul_mode() -> str:
    from tinynerf_amd import models
    return models.MATMUL


def load_first(paths):
    for p in paths:
        try:
            return json.load(open(os.path.join(ROOT, p))), p
        except Exception:       # noqa: BLE001
            continue
    return None, None


def visible_gpus() -> int:
    """GPUs of this node WITHOUT touching HIP or torch.cuda (the launcher parent must stay GPU-free: it starts the ranks):
    KFD topology nodes with SIMDs, filtered by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are set."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return -1                   # unknown (no KFD here): trust --gpus, the children validate
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def bench_grid0(decay: float, res: int = 128):
    """the bench's initial occupancy grid (same construction as oracle/make_psnr_curve.py bench_grid0)"""
    lin = torch.linspace(-1, 1, res)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.where(xx * xx + yy * yy + zz * zz < 0.25, 1.0, decay ** 20).to(torch.float32)


def psnr_replay(o, d, rgbs, dev, n_steps: int, grid0, ho, hd, hrgb):
    """PSNR@step against the reference recipe ON THE CONFIGURATION THE METRIC IS QUOTED ON (BASELINE.json; run.py:53-54,97-319): a second
    trainer with the harness' replayable random streams (TrainConfig.host_shuffle: host ray permutation, counter-RNG jitter, seeded
    refresh jitter) walks the same `n_steps` steps the CPU port of the reference's train() walked for tests/golden/G21_psnr_bench.json
    (oracle/make_psnr_curve.py --bench: same scene, B = S = 1024, same initial grid and parameters, same streams), then renders the held-out
    800 x 800 camera.  Outside every timed window (the timed trainer shuffles on the device).  Returns None when the golden has no entry
    for this step count."""
    from tinynerf_amd.run import TrainConfig, Trainer, psnr as psnr_fn
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "G21_psnr_bench.json")))
    except Exception:       # noqa: BLE001
        return None
    ref = gold["runs"][0]["psnr"].get(str(n_steps))
    if ref is None:
        return {"step": n_steps, "reference": None, "note": "tests/golden/G21_psnr_bench.json holds steps " + ", ".join(sorted(gold["runs"][0]["psnr"], key=int))
                + " (the driver's --warmup 5 --steps 20 ends on 65, the defaults on 70)"}
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=1024, n_samples=1024, seed=0, host_shuffle=True)
    t2 = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev)
    t2.occupancy_grid.grid.copy_(grid0)
    t2.occupancy_grid.mean = float(t2.occupancy_grid.grid.mean().item())
    counts = []
    for _ in range(n_steps):
        counts.append(int(t2.step()["n_samples"]))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        img = t2.render_rays(ho, hd)
    val = float(psnr_fn(img, hrgb))
    gs = gold["runs"][0]["samples_per_step"]
    out = {"step": n_steps, "psnr": val, "reference": ref, "delta_db": val - ref,
           "batch_sizes_equal_first_8": counts[:8] == gs[:8], "loss": t2.loss_value(), "reference_loss": gold["runs"][0]["loss"][n_steps - 1],
           "golden": "tests/golden/G21_psnr_bench.json (oracle/make_psnr_curve.py --bench: CPU port of the reference's train() on this "
                     "configuration, replay streams); gate: |delta_db| < 0.1 (tests/test_hip_psnr.py)"}
    del t2
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--views", type=int, default=20, help="synthetic 800x800 cameras (640k rays each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage rates (profiling runs)")
    ap.add_argument("--cpu-samples", type=int, default=1 << 20, help="packed samples per timed call of the CPU baseline (BASELINE.md: 2^20)")
    ap.add_argument("--full-recipe", action="store_true", help="soak: the reference's whole schedule (8192 steps at B = 1024, LR milestones, "
                    "128 occupancy refreshes) on this workload; prints its own JSON line (time to train, held-out PSNR, step-time extremes)")
    ap.add_argument("--other-steps", type=int, default=8, help="steps per window of the Vanilla / Cobafa side runs")
    ap.add_argument("--other-windows", type=int, default=3)
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each; the first one is the measurement")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank runs the recipe's batch (B rays x S per loader batch, ~B*S samples per rank and step); "
                         "strong = the recipe's B*S samples per step are split over the ranks (PSNR@step comparable with N = 1)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of selected C-ABI entry points on the stream they are launched on."""

    def __init__(self):
        self.records = {}

    def install(self):
        from tinynerf_amd import _lib as L
        orig = L.call
        timer = self

        def timed_call(name, device, *args):
            tag = name
            if name in ("tn_mlp_fwd", "tn_mlp_fwd_stash", "tn_mlp_bwd"):
                desc = args[0]._obj
                tag = name.replace("_stash", "") + (":rgb" if desc.encoding in (L.ENC_DIR_CAT, L.ENC_AUX_CAT) else ":sigma")
            elif name == "tn_kplanes_mlp_bwd_pair":
                tag = name + (":chain" if args[4]._obj.flags & L.MLP_CHAIN_ONLY else ":wgrad")
            if tag not in KERNEL_MODEL or not timer.enabled or (timer.only is not None and tag not in timer.only):
                return orig(name, device, *args)
            if name == "tn_adam_reg_multi":
                rows = sum(int(it.n) for it in args[0])
            else:
                n_arg = {"tn_kplanes_fwd": 3, "tn_kplanes_bwd": 3, "tn_mlp_fwd": 3, "tn_mlp_fwd_stash": 3, "tn_mlp_bwd": 4,
                         "tn_mlp_fwd_stash_pair": 4, "tn_mlp_bwd_pair": 6, "tn_kplanes_mlp_fwd_pair": 6, "tn_kplanes_mlp_bwd_pair": 10}[name]
                rows = int(args[n_arg].value)
            s = torch.cuda.current_stream(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            orig(name, device, *args)
            e1.record(s)
            timer.records.setdefault(tag, []).append((e0, e1, rows))

        self.enabled = False
        self.only = None            # a set of tags: only these are timed (the measured window times the dominant launch alone)
        L.call = timed_call

    def summary(self):
        out = {}
        for tag, recs in self.records.items():
            ms = [e0.elapsed_time(e1) for e0, e1, _ in recs]
            rows = [r for _, _, r in recs]
            out[tag] = dict(launches=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms), avg_rows=sum(rows) / len(rows))
        return out


def physical_cores() -> int:
    """physical cores of the host (unique (package, core) pairs of /proc/cpuinfo); os.cpu_count() when that cannot be read"""
    try:
        pairs, pkg = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                pairs.add((pkg, ln.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(trainer, n_target: int):
    """BASELINE.md's protocol: the CPU port of the hot path on the GPU box's host, N = 2^20 packed samples of the same workload per timed
    call (the first rays of real dynamic batches), 1 warm-up + median of 5, the three stages -- sampler, render forward, render forward +
    loss + backward -- timed separately at ONE stated thread count.  torch's CPU kernels stop scaling well before all hardware threads of
    this host (256 threads ran the port 140 x slower than 32), so the count is min(physical cores, 64)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import tinynerf_oracle as orc
    from oracle import torch_port as tp
    phys, logical = physical_cores(), os.cpu_count() or 1
    threads = max(1, min(phys, 64))
    torch.set_num_threads(threads)
    packs, infos, tgts, n, R = [], [], [], 0, 0
    while n < n_target:                                   # a dynamic batch is ~2^20 samples: one or two of them
        packed, info, target, _ = trainer.build_batch()
        cnt = info[:, 1].long().cumsum(0)
        r = int((cnt <= n_target - n).sum().item())
        if r == 0:
            break
        m = int(cnt[r - 1].item())
        inf = info[:r].clone()
        inf[:, 0] += n
        packs.append(packed[:m].cpu()); infos.append(inf.cpu()); tgts.append(target[:r].cpu())
        n += m; R += r
    packed, info, target = torch.cat(packs), torch.cat(infos), torch.cat(tgts)
    sd = {k: v.detach().cpu().contiguous() for k, v in trainer.renderer.state_dict().items()}
    bg = trainer.renderer.bg_color.cpu() if trainer.renderer.bg_color is not None else None

    def med5(fn):
        fn()                                              # warm-up
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[2]
    t_fb = med5(lambda: tp.grads_of(sd, lambda p: tp.training_loss(p, packed, info, target, bg)))
    with torch.no_grad():
        t_fwd = med5(lambda: tp.render(sd, packed, info, bg))
    # sampler: the numpy restatement of core.py:165-188 on as many loader batches of the bench's rays as give ~N kept samples, the rays
    # dealt to `threads` workers (numpy releases the GIL inside its kernels)
    g = trainer.occupancy_grid
    aabb = np.array([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]], np.float32)
    kw = dict(marcher="aabb", contraction="aabb", grid=g.grid.cpu().numpy(), threshold=float(g.threshold), n_samples=trainer.cfg.n_samples,
              near=0.1, aabb=aabb)
    n_rays = max(1024, int(R))
    pick = torch.randint(0, trainer.rays_o.size(0), (n_rays,), generator=torch.Generator().manual_seed(0)).to(trainer.rays_o.device)
    o_cpu, d_cpu = trainer.rays_o[pick].cpu().numpy(), trainer.rays_d[pick].cpu().numpy()
    chunks = [slice(i, min(i + 256, n_rays)) for i in range(0, n_rays, 256)]
    kept = [0]

    def sample_all():
        with ThreadPoolExecutor(threads) as ex:
            kept[0] = sum(ex.map(lambda sl: orc.ray_provider(o_cpu[sl], d_cpu[sl], **kw)[0].shape[0], chunks))
    t_samp = med5(sample_all)
    stages = {"render_fwd_samples_per_s": n / t_fwd, "render_fwd_bwd_samples_per_s": n / t_fb, "sampler_samples_per_s": kept[0] / t_samp,
              "sampler_candidates_per_s": n_rays * trainer.cfg.n_samples / t_samp, "sampler_rays": n_rays, "sampler_kept_samples": kept[0]}
    return {"value": n / t_fb, "unit": "samples/s", "cores": threads, "kind": "port (render forward + loss + backward; no optimizer step)",
            "threads": threads, "physical_cores": phys, "logical_cpus": logical, "protocol": "BASELINE.md: N = 2^20, 1 warm-up + median of 5 per stage",
            "stages": stages,
            "sample": f"{n} packed samples / {R} rays of the bench's own dynamic batches; torch {torch.__version__} CPU kernels + oracle/weights_ref.c "
                      f"(render), numpy restatement of core.py:165-188 over {threads} worker threads (sampler); every stage at {threads} threads"}


def alloc_counters():
    """device-allocator events so far: a hipMalloc inside a timed window is a 60-240 ms stall"""
    st = torch.cuda.memory_stats()
    return {"device_allocs": int(st.get("num_device_alloc", 0)), "alloc_retries": int(st.get("num_alloc_retries", 0)),
            "segments": int(st.get("segment.all.current", 0)), "reserved_gb": st.get("reserved_bytes.all.current", 0) / 2 ** 30}


def side_profile(method):
    """committed counter pass of a side configuration's step (scripts/pmc_config.sh -> profiles/round6_<method>_pmc_traffic.json): HBM-side bytes
    per step and, per kernel, bytes and duration per launch -> GB/s against the HBM peak"""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", f"round6_{method}_pmc_traffic.json")))
    except Exception:       # noqa: BLE001
        return None
    try:        # the matrix-pipe counters of the same step (scripts/profile_config.sh): busy fraction and the clock the launch ran at
        busy = json.load(open(os.path.join(ROOT, "profiles", f"round6_{method}_mfma_busy.json"))).get("per_kernel", {})
    except Exception:       # noqa: BLE001
        busy = {}
    top = []
    for k, v in list(p.get("kernels", {}).items())[:8]:
        r = {"kernel": k, "bound": "hbm", "traffic": v["bytes_per_launch"], "avg_launch_ms_profiled": (v.get("avg_us_profiled") or 0) / 1e3,
             "achieved": v.get("gbs"), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": v.get("frac_of_hbm_peak"), "launches_per_step": v["launches_per_step"]}
        b = busy.get(k)
        if b and b.get("GRBM_GUI_ACTIVE") and v.get("avg_us_profiled"):
            f = b.get("mfma_pipe_busy_frac", 0.0)
            ghz = b["GRBM_GUI_ACTIVE"] / 8.0 / (v["avg_us_profiled"] * 1e3)
            r["mfma"] = {"pipe_busy_frac": f, "clock_ghz": ghz, "busy_x_clock_over_2p4_ghz": f * ghz / 2.4}
            if f > (r["frac"] or 0.0):
                r["bound"], r["hbm"] = "mfma", {"achieved": r["achieved"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": r["frac"]}
                r["achieved"], r["peak"], r["unit"], r["frac"] = f * ghz / 2.4, 1.0, "fraction of the f16 MFMA peak at 2.4 GHz (pipe busy x clock)", f * ghz / 2.4
        top.append(r)
    return {"hbm_bytes_per_step": p["bytes_per_step"], "rooflines": top, "source": f"profiles/round6_{method}_pmc_traffic.json",
            "mfma_note": "rooflines[].mfma: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of profiles/round6_<method>_mfma_busy.json; clock = "
                         "GRBM_GUI_ACTIVE / 8 / launch duration -- the chip lowers it under matrix load (DESIGN 4.1), so busy x clock / 2.4 GHz is the "
                         "fraction of the guide's MFMA peak; a launch whose pipe-busy fraction exceeds its HBM fraction is labelled bound = mfma"}


def run_other_config(method, o, d, rgbs, tr_main, dev, steps, n_windows, matmul=None, held_out=None):
    """One of the reference's other model configurations on the headline's workload: warm up until the scratch arenas and the
    allocator have stopped growing, then `n_windows` windows of `steps` steps (synchronize on both sides); min / median over
    the windows, every step's time from HIP events on the launch stream, and the allocator's counters over the timed region
    (device_allocs_in_windows must be 0: a fresh hipMalloc in a step is a stall, not kernel time)."""
    from tinynerf_amd import models as tn_models
    from tinynerf_amd.run import TrainConfig, Trainer
    prev_mode = tn_models.MATMUL
    if matmul is not None:
        tn_models.MATMUL = matmul
    mode = tn_models.MATMUL
    c2 = TrainConfig(method=method, scene_type="aabb", batch_size=1024, n_samples=1024, seed=0)
    t2 = Trainer(c2, o, d, rgbs, torch.ones(3, device=dev), dev)
    t2.occupancy_grid.grid.copy_(tr_main.occupancy_grid.grid)
    t2.occupancy_grid.mean = float(t2.occupancy_grid.grid.mean().item())
    t2.occupancy_grid_updates = 10 ** 9                      # the refresh schedule is part of the headline run only
    warm, grown = 0, -1
    while warm < 4 or (warm < 16 and (grown != arenas_grown(t2) or before != alloc_counters()["device_allocs"])):
        grown, before = arenas_grown(t2), alloc_counters()["device_allocs"]
        t2.step()
        torch.cuda.synchronize()
        warm += 1
    a0 = alloc_counters()
    stream = torch.cuda.current_stream(dev)
    win_ms, step_ms, rate = [], [], []
    for _ in range(n_windows):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        evs[0].record(stream)
        n_ = 0.0
        for i in range(steps):
            n_ += t2.step()["n_samples"]
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        t_ = time.perf_counter() - t_
        win_ms.append(t_ / steps * 1e3)
        rate.append(n_ / t_)
        step_ms += [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    a1 = alloc_counters()
    srt = sorted(win_ms)
    best = win_ms.index(srt[0])
    out = {"ms_per_step": srt[len(srt) // 2], "samples_per_s": sorted(rate)[len(rate) // 2], "min_ms_per_step": srt[0],
           "best_samples_per_s": rate[best], "windows_ms_per_step": win_ms, "steps_each": steps, "warmup_steps": warm,
           "step_ms_events": {"min": min(step_ms), "median": sorted(step_ms)[len(step_ms) // 2], "max": max(step_ms)},
           "device_allocs_in_windows": a1["device_allocs"] - a0["device_allocs"], "alloc_retries": a1["alloc_retries"],
           "segments": a1["segments"], "reserved_gb": a1["reserved_gb"], "loss": t2.loss_value()}      # (ms_per_step / samples_per_s: median window)
    # width-256 / 128 stacks: algorithmic FLOP of the whole model (forward + data gradient + weight gradient) over the step
    flop = model_flop_per_sample(t2.renderer)
    if flop:
        tf = flop * out["samples_per_s"] / 1e12
        out["flop_per_sample_step"] = flop
        out["tflops_fp32_equivalent"] = tf
        out["matmul"] = mode
        # TN_MLP_SKIP_LAST: the stack's last Linear(F, F) is merged into the decoders' first layers -- the reference model's FLOP above
        # are what the step is worth, the launches execute 6 F^2 per sample less (one layer's forward, data and weight gradient)
        prod = t2.renderer.__dict__.get("_rows_producer")
        sc = prod.__dict__.get("scratch") if prod is not None else None
        if sc is not None and len(sc) > 2 and sc[2].get("skipped_last"):
            F = prod.params()[-2].size(0)
            out["merged_last_layer"] = {"flop_per_sample_step_executed": flop - 6 * F * F,
                                        "note": "TN_MLP_SKIP_LAST: flop_per_sample_step / tflops_fp32_equivalent count the reference model's "
                                                "products; the merged form skips the last layer's three"}
        if mode == "f16x2":
            # every layer of the stack and the heads' forward as two-term fp16 splits with power-of-two scales (TN_MLP_F16X2): THREE fp16
            # MFMAs per fp32 product block; the heads' backward and the first-layer weight gradients stay on the fp32 MFMA
            out["mfma_frac"] = tf / PEAK_F16X2_TFLOPS
            out["mfma_peak"] = {"tflops": PEAK_F16X2_TFLOPS, "what": "fp16 MFMA peak / 3 (f16x2)"}
            out["vs_fp32_mfma_peak"] = tf / PEAK_FP32_MFMA_TFLOPS
        elif mode == "bf16x3":
            # wide-stack layers on the bf16 matrix cores with exact 3-way operand splits (TN_MLP_BF16X3): fp32-accurate products
            # at six bf16 MFMAs each; the width-64 heads stay on the fp32 MFMA
            out["mfma_frac"] = tf / PEAK_BF16X3_TFLOPS
            out["mfma_peak"] = {"tflops": PEAK_BF16X3_TFLOPS, "what": "bf16 MFMA peak / 6 (bf16x3)"}
            out["vs_fp32_mfma_peak"] = tf / PEAK_FP32_MFMA_TFLOPS
        else:
            out["mfma_frac"] = tf / PEAK_FP32_MFMA_TFLOPS
            out["mfma_peak"] = {"tflops": PEAK_FP32_MFMA_TFLOPS, "what": "fp32 MFMA peak"}
    prof = side_profile(method) if matmul is None else None
    if prof:
        out["hbm_bytes_per_step"] = prof["hbm_bytes_per_step"]
        out["hbm_gbs"] = prof["hbm_bytes_per_step"] / (out["ms_per_step"] * 1e-3) / 1e9
        out["hbm_frac"] = out["hbm_gbs"] / PEAK_HBM_GBS
        out["rooflines"] = prof["rooflines"]
        out["hbm_source"] = prof["source"]
        out["rooflines_note"] = prof["mfma_note"]
    if held_out is not None and matmul is None:
        # an 800 x 800 image through the inference path (run.py:15-50): the wide stacks run as ONE persistent launch per chunk (csrc/mlp_fused_f2.hip)
        import contextlib
        ho, hd = held_out
        with contextlib.redirect_stdout(sys.stderr):
            t2.render_rays(ho, hd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t2.render_rays(ho, hd)
            torch.cuda.synchronize()
        out["render_ms"] = (time.perf_counter() - t0) * 1e3
        if method in ("vanilla", "cobafa"):
            from tinynerf_amd.models import _FusedMLP
            _FusedMLP.layerwise_inference = True
            try:
                with contextlib.redirect_stdout(sys.stderr):
                    t2.render_rays(ho, hd)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    t2.render_rays(ho, hd)
                    torch.cuda.synchronize()
                out["render_ms_layerwise"] = (time.perf_counter() - t0) * 1e3       # (TN_MLP_LAYERWISE: the form of rounds 3 - 5, same image)
            finally:
                _FusedMLP.layerwise_inference = False
    del t2
    tn_models.MATMUL = prev_mode
    return out


def arenas_grown(trainer) -> int:
    g = trainer.scratch.grown if hasattr(trainer, "scratch") else 0
    a = trainer.renderer.__dict__.get("_arena")
    return g + (a.grown if a is not None else 0) + getattr(trainer, "_arena_grown", 0)


def model_flop_per_sample(renderer):
    """2 * MACs of every Linear of the model per sample, x3 for forward + data gradient + weight gradient, minus the data
    gradients nobody needs (first layers whose inputs are encodings of the sample position / ray direction)"""
    lin = [m for m in renderer.modules() if isinstance(m, torch.nn.Linear)]
    fwd = sum(2 * m.in_features * m.out_features for m in lin)
    skip = 0
    fm = renderer.feature_module
    first = next((m for m in fm.modules() if isinstance(m, torch.nn.Linear)), None)
    if first is not None and type(fm).__name__ == "VanillaFeatureMLP":
        skip += 2 * first.in_features * first.out_features            # d / d PE(x): not needed
    cd_first = next((m for m in renderer.rgb_decoder.modules() if isinstance(m, torch.nn.Linear)), None)
    if cd_first is not None:
        skip += 2 * 51 * cd_first.out_features                        # d / d [PE(d), d]: not needed
    return 3 * fwd - skip


def full_recipe(tr, rays, dev, rank, world, sync):
    """Soak / stability record (round-5 verdict, item 8): the reference's WHOLE schedule on the bench's workload -- 2048 * 4096 / B = 8192
    optimizer steps at B = 1024 (run.py:100-103), an occupancy refresh every 64 steps (128 of them), MultiStepLR milestones at 1/2, 3/4, 5/6
    and 9/10 of the schedule (run.py:188-199) -- timed as a whole, with the step-time distribution from HIP events around blocks of 64 steps,
    the learning rate at every milestone and the held-out PSNR at the quarter points.  No CPU golden exists at this length (the CPU port
    needs ~30 s per step); what this shows is that the harness runs the recipe end to end and what it reaches."""
    from tinynerf_amd.run import psnr as psnr_fn
    import contextlib
    ho, hd, hrgb, _, _ = rays.synthetic_scene(n_views=1, res=800, seed=10_007, device=str(dev))
    total = tr.steps
    block = tr.occupancy_grid_updates
    stream = torch.cuda.current_stream(dev)
    curve, lrs, evs, samples = {}, {}, [], 0.0

    def heldout():
        with contextlib.redirect_stdout(sys.stderr), torch.no_grad():
            return float(psnr_fn(tr.render_rays(ho, hd), hrgb))
    curve[0] = heldout()
    sync()
    t0 = time.perf_counter()
    t_eval = 0.0
    for s0 in range(0, total, block):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(min(block, total - s0)):
            samples += tr.step()["n_samples"]
        e1.record(stream)
        evs.append((e0, e1, min(block, total - s0)))
        lrs[tr.train_step] = float(tr.optimizer.param_groups[0]["lr"])
        if tr.train_step in (total // 4, total // 2, 3 * total // 4, total):
            torch.cuda.synchronize()
            te = time.perf_counter()
            curve[tr.train_step] = heldout()
            t_eval += time.perf_counter() - te
    sync()
    dt = time.perf_counter() - t0 - t_eval
    per_block = [e0.elapsed_time(e1) / k for e0, e1, k in evs]
    changes = {}
    prev = None
    for st_, lr in sorted(lrs.items()):
        if lr != prev:
            changes[st_] = lr
        prev = lr
    if rank == 0:
        print(json.dumps({"full_recipe": {"steps": total, "batch_size": tr.cfg.batch_size, "refresh_every": block, "refreshes": len(evs),
                                          "time_to_train_s": dt, "samples": samples, "samples_per_s": samples / dt, "ms_per_step_mean": dt / total * 1e3,
                                          "ms_per_step_blocks_of_%d" % block: {"min": min(per_block), "median": sorted(per_block)[len(per_block) // 2], "max": max(per_block),
                                                                              "note": "each block holds one occupancy refresh"},
                                          "heldout_psnr": {str(k): v for k, v in sorted(curve.items())}, "lr_first_seen_at_step": {str(k): v for k, v in changes.items()},
                                          "milestones": [total // 2, total * 3 // 4, total * 5 // 6, total * 9 // 10], "final_loss": tr.loss_value(),
                                          "occupancy": tr.occupancy_grid.occupancy(), "finite": all(bool(torch.isfinite(p).all()) for p in tr.renderer.parameters()),
                                          "n_gpus": world}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as fresh child processes and
    relay rank 0's JSON line.  The parent never touches the GPU (no call under `torch.cuda`: GPUs are counted from the
    KFD topology files), the children are plain `python bench.py` processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set -- the same environment `python -m torch.distributed.run` gives them.  Any failing rank fails the run."""
    import socket
    import subprocess
    n = args.gpus
    visible = visible_gpus()
    shared = os.environ.get("TN_BENCH_BACKEND", "nccl") != "nccl"
    if 0 <= visible < n and not shared:
        sys.stderr.write(f"bench.py --gpus {n}: only {visible} GPU(s) visible (TN_BENCH_BACKEND=gloo shares GPUs between "
                         f"ranks for debugging; its numbers are not scaling measurements)\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        sys.stderr.write(f"bench.py --gpus {n}: rank exit codes {codes}\n")
        sys.stdout.write(out or "")
        return 1
    # exactly one line on stdout: rank 0's JSON (a backend may chat on stdout, e.g. gloo's "[Gloo] Rank 0 is connected ...")
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    sys.stderr.write("".join(ln + "\n" for ln in (out or "").splitlines() if not ln.startswith("{")))
    sys.stdout.write(lines[-1] + "\n" if lines else "")
    return 0 if lines else 1


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    global torch
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: tinynerf_amd has no CPU path")
    # one rank per GPU over RCCL.  TN_BENCH_BACKEND=gloo (debug) lets several ranks share one GPU to exercise the N > 1 path
    backend = os.environ.get("TN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")      # RCCL's stream ahead of the weight-gradient kernels it overlaps
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    # how many ranks the collective library actually connects (an all-reduce of ones): a scaling record can be checked against it
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer

    timer = KernelTimer()
    timer.install()

    o, d, rgbs, K, _ = rays.synthetic_scene(n_views=args.views, res=800, seed=rank, device=str(dev))
    shard = world if args.scaling == "strong" else 1
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=1024, n_samples=1024, seed=0, shard=shard)
    tr = Trainer(cfg, o, d, rgbs, torch.ones(3, device=dev), dev, rank=rank, world_size=world)
    # occupancy: 1 inside the centred ball of radius 0.5 (normalised coords), decay^20 elsewhere -- built with torch's CPU kernels and
    # uploaded, so that the CPU checker's replay of this configuration (oracle/make_psnr_curve.py --bench -> G21) starts from the same bits
    grid0 = bench_grid0(tr.occupancy_grid.decay).to(dev)
    tr.occupancy_grid.grid.copy_(grid0)
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.full_recipe:
        return full_recipe(tr, rays, dev, rank, world, sync)

    # Event pairs around EVERY modelled launch cost the step ~0.2 ms (ten records per step, each a marker the queue drains to):
    # 3.29 against 3.10 ms on one box.  So the last warm-up steps time all of them to find the dominant launch, the measured window
    # times only that one (two records per step: `roofline`), and the first variance window times the rest (`rooflines`).
    pre = min(args.warmup, 4) if args.windows >= 2 else 0
    for i in range(args.warmup):
        timer.enabled = i >= args.warmup - pre
        tr.step()
    timer.enabled = False
    dominant = None
    if pre:
        torch.cuda.synchronize()
        ps = timer.summary()
        dominant = max(ps, key=lambda t: ps[t]["total_ms"]) if ps else None
        timer.records = {}
        timer.only = {dominant} if dominant else None

    def window(timed: bool):
        """exactly `steps` steps between barrier + synchronize on both sides; max over ranks, samples summed over ranks"""
        sync()
        timer.enabled = timed
        samples, rays_n = 0.0, 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = tr.step()
            samples += st["n_samples"]
            rays_n += st["n_rays"]
        sync()
        dt = time.perf_counter() - t0
        timer.enabled = False
        stats = torch.tensor([dt, samples, rays_n], dtype=torch.float64, device=dev)
        if world > 1:
            tmax = stats[:1].clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(stats[1:])
            stats[0] = tmax[0]
        return stats.tolist()

    a0, g0 = alloc_counters()["device_allocs"], arenas_grown(tr)
    dt, samples, rays_n = window(True)            # THE measurement: `value`, `ms_per_step`, kernel events
    alloc_w0, grown_w0 = alloc_counters()["device_allocs"] - a0, arenas_grown(tr) - g0
    loss = tr.loss_value()
    ks_measured = timer.summary()
    if dominant is not None:
        timer.records, timer.only = {}, None
    extra_windows = [window(dominant is not None and i == 0) for i in range(max(0, args.windows - 1))]      # variance (+ the other launches' events)
    window_ms = [dt / args.steps * 1e3] + [w[0] / args.steps * 1e3 for w in extra_windows]

    # PSNR@step (the second half of BASELINE.json's metric, run.py:53-54): one held-out 800x800 view of the same synthetic
    # scene -- a camera no rank trains on -- rendered with the parameters as they are after the timed steps (every rank holds the
    # same parameters; rank 0 renders)
    psnr_at_step = None
    ho = hd = hrgb = None
    if rank == 0:
        from tinynerf_amd.run import psnr as psnr_fn
        ho, hd, hrgb, _, _ = rays.synthetic_scene(n_views=1, res=800, seed=10_007, device=str(dev))
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # (chunks of pure background print the reference's "Empty iteration" line:
            tr.render_rays(ho, hd)                        # core.py:253 -- stdout carries the JSON line only); warm-up: scratch arenas
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img = tr.render_rays(ho, hd)
            torch.cuda.synchronize()
            render_ms = (time.perf_counter() - t0) * 1e3       # (before the PSNR: its first torch ops load their code objects, 30 - 170 ms)
        psnr_at_step = {"step": tr.train_step, "psnr": float(psnr_fn(img, hrgb)), "render_ms": render_ms,
                        "view": "held-out 800x800 camera (rays.synthetic_scene(n_views=1, seed=10007)), inference path (training=False sampling)",
                        "samples_per_step": samples / args.steps,
                        "note": "psnr = the TIMED trainer (device-side shuffle: its ray order is its own); `replay` = a second trainer on the "
                                "replayable streams, held against the CPU port of the reference's train() on this very configuration"}
        del img
        if world == 1 and args.views == 20:
            try:
                psnr_at_step["replay"] = psnr_replay(o, d, rgbs, dev, tr.train_step, grid0, ho, hd, hrgb)
            except Exception as e:                                  # noqa: BLE001 -- the headline line must still be printed
                psnr_at_step["replay"] = {"error": repr(e)}
    # the occupancy refresh (run.py:248-249) runs every 16 * 4096 / B steps: a window of K < 64 steps behind a short warm-up
    # never contains one, so its cost is measured here and folded into `value_with_refresh` at its amortised weight
    refresh = None
    if rank == 0 or world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        grid_keep, mean_keep = tr.occupancy_grid.grid.clone(), tr.occupancy_grid.mean
        torch.cuda.synchronize()
        e0.record(torch.cuda.current_stream(dev))
        tr.occupancy_grid.update(tr.sigma_fn, seed=12345)
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        tr.occupancy_grid.grid.copy_(grid_keep)                      # (the stage timings below run on the bench's own grid)
        tr.occupancy_grid.mean = mean_keep
        ms = e0.elapsed_time(e1)
        refresh = {"ms_per_refresh": ms, "every_steps": tr.occupancy_grid_updates, "ms_per_step_amortised": ms / tr.occupancy_grid_updates,
                   "in_window0": any((args.warmup + i) % tr.occupancy_grid_updates == 0 for i in range(args.steps))}
        del grid_keep

    # stage rates on one batch of the same workload (sampler / render fwd / render fwd+bwd)
    stages = None
    if rank == 0 and world == 1 and not args.no_stages:      # (N = 1 only: a one-rank backward would start collectives nobody joins)
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps
        packed, info, target, _ = tr.build_batch()
        packed, info, target = packed.clone(), info.clone(), target.clone()    # build_batch hands out views of reused buffers
        nb = packed.size(0)
        t_samp = timed(lambda: tr.build_batch())
        with torch.no_grad():
            t_fwd = timed(lambda: tr.renderer(packed, info))
        def fwd_bwd():
            tr.optimizer.zero_grad(set_to_none=False)
            torch.nn.functional.mse_loss(tr.renderer(packed, info), target).backward()
        t_fb = timed(fwd_bwd)
        stages = {"sampler_samples_per_s": nb / t_samp, "render_fwd_samples_per_s": nb / t_fwd,
                  "render_fwd_bwd_samples_per_s": nb / t_fb, "batch_samples": nb,
                  # algorithmic FLOP of both heads over the stage time (the inference forward evaluates the colour head where
                  # w > 0: every sample of this batch, random-init sigma never terminates a ray) against the fp32 MFMA peak
                  "render_fwd_mfma_frac": FLOP_HEADS * nb / t_fwd / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                  "render_fwd_bwd_mfma_frac": FLOP_STEP * nb / t_fb / 1e12 / PEAK_FP32_MFMA_TFLOPS}

    # the reference's other two model configurations on the same workload (BASELINE configs 2 and 5), N = 1 only, reported
    # beside the headline (never part of `value`)
    others = None
    if rank == 0 and world == 1 and not args.no_stages:
        others = {}
        for key, method, matmul in (("kplanes_fp32_heads", "kplanes", "fp32"), ("vanilla", "vanilla", None), ("cobafa", "cobafa", None),
                                    ("vanilla_bf16x3", "vanilla", "bf16x3"), ("vanilla_fp32_mfma", "vanilla", "fp32")):
            try:
                others[key] = run_other_config(method, o, d, rgbs, tr, dev, args.other_steps, args.other_windows, matmul,
                                               held_out=(ho, hd) if ho is not None else None)
            except Exception as e:                                      # noqa: BLE001 -- the headline line must still be printed
                others[key] = {"error": repr(e)}
            # a Trainer and its renderer reference each other (the batch hints hold bound methods): the side run's arenas -- 11 GB of Vanilla
            # workspace -- are only released once the cycle collector has run (round 5 kept 78 GB reserved after the side runs)
            import gc
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0:
        ks = timer.summary()              # every modelled launch (window 1 when the measured window timed the dominant one alone) ...
        ks.update(ks_measured)            # ... and the dominant launch from the measured window itself
        pmc, PMC_PROFILE = load_first(PMC_PROFILES)
        mfma, MFMA_PROFILE = load_first(MFMA_PROFILES)

        def mfma_busy_of(tag):
            """matrix-pipe busy fraction of the tag's kernels from the committed counter pass: sum of MFMA busy cycles over
            sum of (GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- what the MFMA roofline fraction must agree with"""
            if not mfma or tag not in MFMA_KERNELS:
                return None
            busy = act = 0.0
            for k, v in mfma.get("per_kernel", mfma).items():     # (either the wrapped or the raw output of scripts/pmc_mfma.sh)
                if not isinstance(v, dict):
                    continue
                if any(k.startswith(pref) for pref in MFMA_KERNELS[tag]):
                    busy += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
                    act += v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
            return busy / act if act else None

        from tinynerf_amd import fused as _fused
        from tinynerf_amd import run as _run_mod
        mode = _matmul_mode()
        lean = bool(_fused.KP_LEAN and mode == "f16x2")

        def roofline_of(tag):
            bound, unit_work, unit, kernels = KERNEL_MODEL[tag]
            k = ks[tag]
            sec = k["avg_ms"] * 1e-3
            traffic = pmc.get("per_entry", {}).get(tag) if pmc else None
            cls, limit = launch_class(tag, mode, lean)
            r = {"kernel": tag, "kernels": kernels, "avg_launch_ms": k["avg_ms"], "ms_per_step": k["total_ms"] / args.steps,
                 **({"shares_the_chip": "TN_ADAM_OVERLAP: tn_adam_reg_multi (side stream) runs beside the first weight-gradient launch; serial: "
                                        "0.22 ms / 0.83 ms"} if (_run_mod.ADAM_OVERLAP and world == 1 and tag in ("tn_adam_reg_multi", "tn_kplanes_mlp_bwd_pair:wgrad")) else {}),
                 "rows_per_launch": k["avg_rows"], "row": unit, "algorithmic_per_row": unit_work, "traffic": traffic,
                 "traffic_source": PMC_PROFILE if traffic is not None else None, "instructions": cls, "limited_by": limit}
            tflops = unit_work * k["avg_rows"] / sec / 1e12 if bound in ("mfma", "atomic") and unit_work else None
            if tflops:
                # `frac` is quoted against the dense peak of the instruction class the launch issues (MFMA_CLASS); the fp32-equivalent
                # rate over the fp32 MFMA peak is given beside it and means "how much faster than an fp32-MFMA kernel could be"
                peak = MFMA_CLASS[cls][1] if cls in MFMA_CLASS else PEAK_FP32_MFMA_TFLOPS
                r.update(bound="mfma", achieved=tflops, peak=peak, unit="TFLOP/s (fp32-equivalent)", frac=tflops / peak,
                         vs_fp32_mfma_peak=tflops / PEAK_FP32_MFMA_TFLOPS)
            elif bound == "hbm":
                gbs = unit_work * k["avg_rows"] / sec / 1e9
                r.update(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS)
            else:               # the stand-alone scatter has no matrix work: HBM-side bytes of the PMC pass against the HBM peak
                gbs = traffic / sec / 1e9 if traffic else None
                r.update(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS if gbs else None)
            if bound == "atomic":   # memory-side fp32 atomics: lane-atomics per launch from the PMC pass (WRITE_SIZE counts 4 B per lane-atomic)
                lanes = pmc.get("lane_atomics_per_entry", {}).get(tag) if pmc else None
                ach = lanes / sec / 1e9 if lanes else None
                if ach:
                    # the launch is limited by the atomic path, not by the matrix pipe: `bound` says so, the matrix-pipe view moves to `mfma`
                    r["mfma"] = {k: r[k] for k in ("achieved", "peak", "unit", "frac", "vs_fp32_mfma_peak") if k in r}
                    r.update(bound="atomic", achieved=ach, peak=PEAK_ATOMIC_GLANES, unit="G lane-atomics/s", frac=ach / PEAK_ATOMIC_GLANES,
                             lane_atomics_per_launch=lanes,
                             peak_source="scripts/microbench/atomic_scaling.hip (this repo's measurement: 325 G lane-atomics/s on random full lines from "
                                         ">= 4096 waves; not a figure of MI355X_MICROARCH.md; history and conditions: DESIGN 4)")
                    r.pop("vs_fp32_mfma_peak", None)
            if traffic is not None:
                r["hbm_gbs"] = traffic / sec / 1e9
            try:
                busy = mfma_busy_of(tag)
            except Exception:        # noqa: BLE001 -- an odd profile file must not cost the bench line
                busy = None
            if busy is not None:
                r["mfma_busy"] = {"frac": busy, "source": MFMA_PROFILE}      # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
            return r

        step_ms = dt / args.steps * 1e3
        # every timed launch that is >= 5 % of the step, largest first; `roofline` = the dominant one
        roofs = [roofline_of(t) for t in sorted(ks, key=lambda t: -ks[t]["total_ms"]) if ks[t]["total_ms"] / args.steps >= 0.05 * step_ms]
        roof = roofs[0] if roofs else None
        per_gpu_samples = samples / args.steps / world
        whole = {"mfma_tflops": FLOP_STEP * per_gpu_samples / (step_ms * 1e-3) / 1e12,
                 "mfma_frac": FLOP_STEP * per_gpu_samples / (step_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                 "note": "algorithmic FLOP of both heads (forward 56 192 + data gradient 49 664 + weight gradient 56 192 per sample) over the "
                         "whole step, fp32-equivalent, against the fp32 MFMA peak"}
        if pmc and pmc.get("bytes_per_step"):
            whole.update(hbm_bytes_per_step=pmc["bytes_per_step"], hbm_gbs=pmc["bytes_per_step"] / (step_ms * 1e-3) / 1e9,
                         hbm_frac=pmc["bytes_per_step"] / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, hbm_source=PMC_PROFILE)
        srt = sorted(window_ms)
        if stages:
            tf = FLOP_HEADS * stages["batch_samples"] / (stages["batch_samples"] / stages["render_fwd_samples_per_s"]) / 1e12
            cls = "f16x2" if mode == "f16x2" else "fp32"
            stages["render_fwd_roofline"] = {"tflops_fp32_equivalent": tf, "instructions": cls, "frac_of_instruction_class_peak": tf / MFMA_CLASS[cls][1],
                                             "vs_fp32_mfma_peak": tf / PEAK_FP32_MFMA_TFLOPS}
        # <= 1 KB: what a reader of the driver's record needs first (repeated as the LAST key: a tail of the line keeps it as well)
        oc = others or {}
        summary = {"value": samples / dt, "ms_per_step": step_ms, "windows_median_ms": srt[len(srt) // 2], "matmul": mode, "lean": lean,
                   "kernels_ms": {t.replace("tn_kplanes_mlp_", "").replace("tn_", ""): round(v["total_ms"] / args.steps, 4) for t, v in sorted(ks.items())},
                   "other_ms": {k: round(v["ms_per_step"], 3) for k, v in oc.items() if isinstance(v, dict) and "ms_per_step" in v},
                   "other_render_ms": {k: round(v["render_ms"], 1) for k, v in oc.items() if isinstance(v, dict) and "render_ms" in v},
                   "render_fwd_samples_per_s": stages["render_fwd_samples_per_s"] if stages else None,
                   "sampler_samples_per_s": stages["sampler_samples_per_s"] if stages else None,
                   "psnr": ({"step": psnr_at_step["step"], "timed_run": round(psnr_at_step["psnr"], 3),
                             **({k: (round(v, 4) if isinstance(v, float) else v) for k, v in (psnr_at_step.get("replay") or {}).items()
                                 if k in ("psnr", "reference", "delta_db")})} if psnr_at_step else None),
                   "hbm_bytes_per_step": whole.get("hbm_bytes_per_step"), "dominant": roof["kernel"] if roof else None,
                   "dominant_bound": roof["bound"] if roof else None, "dominant_frac": roof["frac"] if roof else None,
                   # the same step with every product on v_mfma_f32_32x32x2_f32 (other_configs.kplanes_fp32_heads): the conservative figure
                   "value_fp32_mfma": (oc.get("kplanes_fp32_heads") or {}).get("samples_per_s"),
                   "ms_per_step_fp32_mfma": (oc.get("kplanes_fp32_heads") or {}).get("ms_per_step"),
                   "ranks_seen": ranks_seen}
        line = {
            "summary": summary,
            "metric": "ray-samples/sec (K-Planes training step: sampler + render fwd + bwd + Adam)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32 (f16x2-split products)" if mode == "f16x2" else ("f32 (bf16x3-split products)" if mode == "bf16x3" else "f32"), "data": "synthetic",
            "ranks_seen": ranks_seen,
            "config": {"workload": "K-Planes Lego-shaped 800x800, aabb, B=1024 rays x S=1024, dynamic batches of ~2^20 packed samples, 128^3 occupancy ball",
                       # fp32 storage, accumulation and results everywhere; what the matrix products run on (parity tests at the fp32 tolerances):
                       "matmul": {"f16x2": "heads' forward (+ its rebuild inside the weight-gradient launch, TN_MLP_LEAN) as two-term fp16 splits with power-of-two "
                                           "scales on v_mfma_f32_32x32x16_f16; weight-gradient products G x H as exact three-term bf16 splits on "
                                           "v_mfma_f32_32x32x16_bf16; data-gradient chain and first-layer weight gradients on v_mfma_f32_32x32x2_f32",
                                  "fp32": "every product on v_mfma_f32_32x32x2_f32", "bf16x3": "heads as fp32; wide stacks (other_configs) as exact bf16 triplets"}[mode],
                       "TN_MATMUL": mode, "TN_KP_LEAN": lean,
                       # N == 1: the planes' optimizer pass runs on a stream of its own beside the weight-gradient launches -- the per-launch times
                       # of tn_adam_reg_multi and tn_kplanes_mlp_bwd_pair:wgrad below are times SHARING the chip (their sum is not step time)
                       "TN_ADAM_OVERLAP": bool(_run_mod.ADAM_OVERLAP and world == 1),
                       "parallelism": (f"dp{world} ({args.scaling} scaling: " + ("every rank runs the recipe's batch" if args.scaling == "weak" else
                                                                                 f"the recipe's B*S samples per step split over the ranks, {1024 // world} rays per loader batch and rank")
                                       + f"): rays sharded over {world} ranks (one per GPU), RCCL all-reduce of plane / MLP gradients"
                                       + ("" if backend == "nccl" else f" [DEBUG backend {backend}: ranks share GPUs, not a scaling number]"))
                                      if world > 1 else "single GPU",
                       "samples_per_step_per_gpu": samples / args.steps / world, "rays_per_step_per_gpu": rays_n / args.steps / world},
            "loss": loss,
            "psnr_at_step": psnr_at_step,
            "refresh": refresh,
            # `value` is exactly K steps as timed; with the refresh at its amortised share (unless the window already held one)
            "value_with_refresh": (samples / (dt + (0.0 if refresh["in_window0"] else args.steps * refresh["ms_per_step_amortised"] * 1e-3))
                                   if refresh else None),
            "stages": stages,
            "other_configs": others,
            "kernel_timing": {"measured_window": sorted(ks_measured), "window_1": sorted(t for t in ks if t not in ks_measured),
                              "note": "HIP-event pairs on the launch stream: window 0 around the dominant launch only, window 1 around every modelled launch"},
            "windows": {"n": len(window_ms), "steps_each": args.steps, "ms_per_step": window_ms, "min": srt[0], "median": srt[len(srt) // 2],
                        "note": "window 0 = value; window 1 carries all launch events; the window that contains step 64 the occupancy refresh"},
            "kernels_ms_per_step": {t: v["total_ms"] / args.steps for t, v in sorted(ks.items())},
            "allocator": dict(alloc_counters(), device_allocs_in_window0=alloc_w0, arenas_grown_in_window0=grown_w0),
            "roofline": roof,
            "rooflines": roofs,
            "whole_step": whole,
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (the other ranks would sit in the closing barrier)
            line["cpu_baseline"] = cpu_baseline(tr, args.cpu_samples)
        line["summary_tail"] = summary
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
