"""PSNR@step -- the second half of BASELINE.json's metric -- of the HIP Trainer against the CPU port of the reference's train().

Both goldens come from ``oracle/make_psnr_curve.py`` (build container): ``oracle/torch_port.reference_training`` -- the loop as the
reference runs it (run.py:97-319): shuffled loader stream, ``t += U * delta`` sampling jitter, jittered occupancy refreshes every 64
steps, Adam + MultiStepLR, the never-unscaled 2^10 loss scale -- on the synthetic scene (20 training views + 1 held-out view at
100 x 100, B = 1024, S = 128, 128^3 occupancy grid, full 128 / 256 / 512 K-Planes), 300 steps, held-out PSNR (run.py:53-54) every 50.

* **G18 (replay): the gate.**  The port takes every random choice from the streams the HIP harness defines (host permutation of the
  rays, counter RNG for the sampling and refresh jitter, restated in the oracle and pinned bit for bit by tests/test_hip_core.py and
  tests/test_hip_training.py).  The production path (side-stream sampler prefetch, fused gather / scatter kernels, TV folded into
  Adam, device-side batch rule) then walks the SAME rays with the SAME jitter from the SAME initial parameters: what is left is fp32
  summation order, amplified by Adam.  PSNR at equal step count must agree within 0.1 dB (north star) at every evaluated step
  up to 100 (10, 20, ..., 100; measured: <= 0.001 dB up to step 50, 0.02 dB at 100), the first steps' dynamic batches must have the
  same size and their losses agree to 1e-4.  Beyond ~125 steps this recipe is chaotic on this scene: the held-out PSNR oscillates by +- 1 dB at lr 1e-2 and ANY two fp32
  evaluations of the same trajectory decorrelate -- shown here by the control: the HIP trainer run twice with identical seeds
  (its plane-gradient atomics arrive in a different order, nothing else differs) diverges from itself over the same horizon.
  At 200 - 300 steps a single curve can therefore only be held inside the recipe's own seed-to-seed envelope (3 x the spread over
  the G17 seeds); the statistical statement about those steps is G17's.
* **G17 (independent streams): a cross-check.**  The port draws from its own numpy generator, as a second machine running the
  reference would.  Single curves then differ by the recipe's seed-to-seed noise (0.2 dB at 50 steps, 0.7 - 1 dB later: measured
  here over 16 seeds), so seed means are compared, within 2.5 standard errors of their difference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHECK_AT = (50, 100, 200, 300)
TIGHT_UNTIL = 100                # shared random streams: 0.1 dB at every evaluated step up to here; later steps: see the module docstring


def _scene(c):
    from tinynerf_amd import rays
    o, d, rgbs, _, _ = rays.synthetic_scene(n_views=c["n_views"], res=c["res"], seed=c["scene_seed"], device=DEV)
    n_train = (c["n_views"] - 1) * c["res"] ** 2
    return (o[:n_train].contiguous(), d[:n_train].contiguous(), rgbs[:n_train].contiguous()), (o[n_train:], d[n_train:], rgbs[n_train:])


def _run(seed, c, train, held, eval_at, host_shuffle):
    from tinynerf_amd.run import TrainConfig, Trainer, psnr
    dev = torch.device(DEV)
    cfg = TrainConfig(method=c["method"], scene_type="aabb", batch_size=c["batch_size"], n_samples=c["n_samples"], seed=seed,
                      occupancy_res=c["occupancy_res"], host_shuffle=host_shuffle)
    tr = Trainer(cfg, *train, torch.ones(3, device=dev), dev)
    if "lr" in c:                                   # (the deep stacks' goldens: run.py:110's 1e-2 kills them on this scene, port and GPU alike)
        for grp in tr.optimizer.param_groups:
            grp["lr"] = grp["initial_lr"] = c["lr"]
        tr.scheduler.base_lrs = [c["lr"] for _ in tr.scheduler.base_lrs]
    if c["method"] == "cobafa":
        tr.renderer.feature_module.dropout.p = 0.0  # (the process RNG cannot be shared with the port: off on both sides)
    curve, losses, counts = {}, [], []
    for step in range(max(eval_at) + 1):
        if step in eval_at:
            with torch.no_grad():
                curve[step] = float(psnr(tr.render_rays(held[0], held[1]), held[2]))
        if step < max(eval_at):
            st = tr.step()
            counts.append(int(st["n_samples"]))
            if step < 16:
                losses.append(tr.loss_value())
    assert all(torch.isfinite(p).all().item() for p in tr.renderer.parameters())
    return curve, losses, counts


def test_psnr_at_step_on_the_reference_trajectory():
    """G18: same rays, same jitter, same initial parameters -> PSNR@step within 0.1 dB of the CPU port of the reference for as
    long as two fp32 evaluations of this trajectory stay together at all (control: HIP against HIP)."""
    g = json.load(open(os.path.join(GOLDEN, "G18_psnr_replay.json")))
    g17 = json.load(open(os.path.join(GOLDEN, "G17_psnr_curve.json")))
    assert g["replay"] and g["steps"] >= 300
    c, eval_at = g["config"], [int(e) for e in g["eval_at"]]
    tight_at = [s for s in eval_at if 0 < s <= TIGHT_UNTIL]
    assert len(tight_at) >= 4 and all(s in eval_at for s in (50, 100, 200, 300))
    envelope = {s: 3.0 * float(np.std([run["psnr"][str(s)] for run in g17["runs"]], ddof=1)) for s in eval_at if s >= 200}
    train, held = _scene(c)
    report, control = {}, {}
    for run in g["runs"]:
        curve, losses, counts = _run(run["seed"], c, train, held, eval_at, host_shuffle=True)
        twin = _run(run["seed"], c, train, held, eval_at, host_shuffle=True)[0]          # the same run again: atomics order only
        # the first refresh-free stretch: the same grid on both sides, so bit-identical rays + jitter give the same batch sizes
        assert counts[:8] == run["samples_per_step"][:8], (counts[:8], run["samples_per_step"][:8])
        np.testing.assert_allclose(losses[:8], run["loss"][:8], rtol=1e-4)
        report[run["seed"]] = {s: (round(curve[s], 3), round(run["psnr"][str(s)], 3)) for s in eval_at}
        control[run["seed"]] = {s: round(abs(curve[s] - twin[s]), 3) for s in eval_at}
        assert abs(curve[0] - run["psnr"]["0"]) < 2e-3, report[run["seed"]]
        for s in tight_at:
            # 0.1 dB (north star) -- or, where this very trajectory has already begun to part from ITSELF (the twin: same seeds, only the
            # order of the plane-gradient atomics differs), three times that distance: no comparison can be tighter than the run is determined
            assert abs(curve[s] - run["psnr"][str(s)]) < max(0.1, 3.0 * abs(curve[s] - twin[s])), (run["seed"], s, report[run["seed"]], control[run["seed"]])
        for s, env in envelope.items():
            assert abs(curve[s] - run["psnr"][str(s)]) <= max(env, 0.1), (run["seed"], s, env, report[run["seed"]])
    print("PSNR@step (HIP, CPU port of the reference) per seed:", report)
    print("control |HIP - HIP| with identical seeds per seed:", control)
    # and it learns -- judged where the trajectory is still determined (by step 300 this 100 x 100 run has drifted back to 12 - 14 dB on
    # both sides, 12.6 / 13.2 in the port: a +3 dB bar at step 300 failed one run in three on chaos alone)
    assert all(v[100][0] > v[0][0] + 3.0 for v in report.values())


def test_psnr_at_step_seed_means_with_independent_streams():
    """G17: no shared random stream -- seed means within 2.5 standard errors of their difference; 16 seeds here."""
    g = json.load(open(os.path.join(GOLDEN, "G17_psnr_curve.json")))
    assert not g.get("replay") and g["steps"] >= 300 and len(g["runs"]) >= 3
    c, eval_at = g["config"], [int(e) for e in g["eval_at"]]
    train, held = _scene(c)
    ref = np.array([[run["psnr"][str(s)] for s in eval_at] for run in g["runs"]])
    got = np.array([[_run(seed, c, train, held, eval_at, host_shuffle=False)[0][s] for s in eval_at] for seed in range(16)])
    report = {}
    for s in CHECK_AT:
        i = eval_at.index(s)
        se = float(np.sqrt(ref[:, i].var(ddof=1) / ref.shape[0] + got[:, i].var(ddof=1) / got.shape[0]))
        report[s] = (round(float(got[:, i].mean()), 3), round(float(ref[:, i].mean()), 3), round(se, 3))
    print("PSNR@step seed means (HIP over 16 seeds, CPU port over %d seeds, standard error of the difference):" % ref.shape[0], report)
    for s, (m_got, m_ref, se) in report.items():
        assert abs(m_got - m_ref) <= 2.5 * max(se, 0.05), (s, report)


@pytest.mark.parametrize("golden,tight_until", [("G19_psnr_replay_vanilla", 80), ("G20_psnr_replay_cobafa", 150)])
def test_psnr_at_step_on_the_reference_trajectory_deep_stacks(golden, tight_until, matmul):
    """G19 / G20: the same replay comparison for the other two model configurations (Vanilla 256 x 10 -- BASELINE config 2 -- and
    Cobafa, config 5's model), in both matrix modes, at lr 1e-3 (see make_psnr_curve.py).  Cobafa (round 6: regenerated from the
    reference's constructors, G22): PSNR at equal step count within 0.1 dB of the CPU port at every evaluated step through 150 (measured:
    <= 0.005 dB); at step 200 the curve climbs 0.024 dB per step and the three matrix modes sit + 0.07 / + 0.07 / - 0.03 dB around the
    port's 12.36 -- a lag of one to three optimizer steps, inside the envelope the later steps of every curve are held to.  Vanilla: within 0.1 dB
    through step 80 (measured: <= 0.014 dB); at step 90 the stack leaves its plateau (12.3 -> 14.9 dB in ten steps, 0.27 dB per
    step) and the HIP runs -- bf16x3 twice and fp32 MFMA: 14.77 / 14.78 / 14.71 -- sit 0.2 dB, i.e. less than one optimizer step,
    behind the port's 14.95; from there the curves wander +- 0.4 dB around each other like the K-Planes ones (1.5 dB envelope)."""
    g = json.load(open(os.path.join(GOLDEN, golden + ".json")))
    assert g["replay"] and g["steps"] >= 200 and g["config"]["lr"] == 1e-3
    c, eval_at = g["config"], [int(e) for e in g["eval_at"]]
    train, held = _scene(c)
    run = g["runs"][0]
    curve, losses, counts = _run(run["seed"], c, train, held, eval_at, host_shuffle=True)
    report = {s_: (round(curve[s_], 3), round(run["psnr"][str(s_)], 3)) for s_ in eval_at}
    print(golden, matmul, "PSNR@step (HIP, CPU port):", report)
    assert counts[:8] == run["samples_per_step"][:8], (counts[:8], run["samples_per_step"][:8])
    np.testing.assert_allclose(losses[:8], run["loss"][:8], rtol=2e-4)
    for s_ in eval_at:
        assert abs(curve[s_] - run["psnr"][str(s_)]) < (2e-3 if s_ == 0 else 0.1 if s_ <= tight_until else 1.5), (s_, report)
    assert curve[max(eval_at)] > curve[0] + 1.0, report


@pytest.mark.timeout(900)
@pytest.mark.parametrize("heads_mode", ["f16x2", "fp32"])
def test_psnr_at_step_on_the_headline_configuration(heads_mode, monkeypatch):
    """G21 (round 5): PSNR@step pinned on the configuration BASELINE.json's metric is quoted on -- bench.py's 20 synthetic 800 x 800
    cameras, B = 1024 rays x S = 1024, its occupancy ball, seed 0 -- against the CPU port of the reference's train() on the same replayable
    streams (oracle/make_psnr_curve.py --bench, ~1 h of CPU).  After the driver's 5 + 3 x 20 = 65 steps (across the occupancy refresh at
    step 64) the held-out 800 x 800 view must agree within the north star's 0.1 dB, in both head forms (f16x2: the default, with the lean
    backward of round 5; fp32: the stash form on the fp32 MFMA); the first dynamic batches must have the golden's sizes and the loss of
    the last step agree to 2 %.  bench.py prints the same comparison as psnr_at_step.replay."""
    import importlib.util
    from tinynerf_amd import models, rays
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.torch = torch
    monkeypatch.setattr(models, "MATMUL", heads_mode)
    gold = json.load(open(os.path.join(GOLDEN, "G21_psnr_bench.json")))
    c = gold["config"]
    assert c["n_views"] == 20 and c["res"] == 800 and c["batch_size"] == 1024 and c["n_samples"] == 1024 and gold["replay"]
    dev = torch.device(DEV)
    o, d, rgbs, _, _ = rays.synthetic_scene(n_views=c["n_views"], res=c["res"], seed=c["scene_seed"], device=DEV)
    ho, hd, hrgb, _, _ = rays.synthetic_scene(n_views=1, res=c["res"], seed=c["heldout_seed"], device=DEV)
    grid0 = bench.bench_grid0(0.01 ** (1 / 16)).to(dev)
    res = bench.psnr_replay(o, d, rgbs, dev, 65, grid0, ho, hd, hrgb)
    assert res["reference"] == gold["runs"][0]["psnr"]["65"] and res["batch_sizes_equal_first_8"], res
    assert abs(res["delta_db"]) < 0.1, res
    assert abs(res["loss"] - res["reference_loss"]) <= 0.02 * res["reference_loss"], res
