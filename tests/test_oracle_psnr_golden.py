"""CPU: the committed PSNR@step goldens (G17 / G18, oracle/make_psnr_curve.py) are reproducible from the committed oracle -- the
first steps of the replay run (same scene, same initial parameters, the harness-defined random streams restated in the port) give
the golden's batch sizes, losses and step-0 PSNR again.  (The GPU side of these goldens: tests/test_hip_psnr.py.)"""
import importlib.util
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _maker():
    spec = importlib.util.spec_from_file_location("make_psnr_curve", os.path.join(ROOT, "oracle", "make_psnr_curve.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_replay_golden_is_reproducible_from_the_oracle():
    from oracle import tinynerf_oracle as orc
    from oracle import torch_port as tp
    mk = _maker()
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "G18_psnr_replay.json")))
    assert g["replay"] and g["config"] == mk.CONFIG and g["eval_at"] == [e for e in mk.EVAL_AT_REPLAY if e <= g["steps"]]
    run = next(r for r in g["runs"] if r["seed"] == 0)
    (o, d, rgbs), (ho, hd, hrgb) = mk.scene()
    c = mk.CONFIG
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    seen = {}

    def eval_fn(step, sd, grid, thr):
        packed, info = orc.ray_provider(ho.numpy(), hd.numpy(), marcher="aabb", contraction="aabb", grid=grid, threshold=thr,
                                        n_samples=c["n_samples"], near=0.1, aabb=aabb)
        with torch.no_grad():
            img = tp.render({k: v.detach() for k, v in sd.items()}, torch.from_numpy(packed), torch.from_numpy(info), torch.ones(3))
        seen[step] = float(-10.0 * torch.log10(torch.mean((img - hrgb) ** 2)))
    n_steps = 2
    losses, _, counts = tp.reference_training(mk.initial_state(0), o.numpy(), d.numpy(), rgbs.numpy(), method=c["method"],
                                              batch_size=c["batch_size"], n_samples=c["n_samples"], n_steps=n_steps,
                                              occupancy_res=c["occupancy_res"], replay={"seed": 0, "rank": 0}, eval_at=[0], eval_fn=eval_fn)
    assert [cn[0] for cn in counts] == run["samples_per_step"][:n_steps] and [cn[1] for cn in counts] == run["rays_per_step"][:n_steps]
    np.testing.assert_allclose(losses, run["loss"][:n_steps], rtol=1e-4)          # (another host: another expf / GEMM blocking)
    assert abs(seen[0] - run["psnr"]["0"]) < 1e-3


def test_independent_stream_golden_is_well_formed():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "G17_psnr_curve.json")))
    mk = _maker()
    assert not g.get("replay") and g["config"] == mk.CONFIG and len(g["runs"]) >= 3 and g["steps"] >= 300
    for run in g["runs"]:
        assert len(run["loss"]) == g["steps"] and all(np.isfinite(run["loss"]))
        p = run["psnr"]
        assert 8.5 < p["0"] < 10.0 and p["300"] > p["0"] + 3.0                  # the recipe learns on this scene


def test_headline_configuration_golden_is_well_formed():
    """G21 (oracle/make_psnr_curve.py --bench): the replay run on bench.py's own configuration -- the step counts the bench ends on are
    there, the recipe learns, and the batch rule lands on ~2^20 samples per step."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "G21_psnr_bench.json")))
    mk = _maker()
    assert g["replay"] and g["config"] == mk.BENCH_CONFIG and g["steps"] == 70 and g["eval_at"] == list(mk.BENCH_EVAL_AT)
    run = g["runs"][0]
    p = run["psnr"]
    assert set(p) == {"0", "30", "65", "70"} and p["0"] < p["30"] < p["65"] < p["70"] and 19.0 < p["65"] < 21.0
    assert len(run["loss"]) == 70 and all(np.isfinite(run["loss"])) and run["loss"][-1] < 0.25 * run["loss"][0]
    assert all(0.95 * 2 ** 20 < n < 1.05 * 2 ** 20 for n in run["samples_per_step"])
    # the initial grid of the golden is bench.py's: the same construction on both sides
    grid = mk.bench_grid0(128)
    assert grid.dtype == np.float32 and abs(float(grid.mean()) - (0.0654498 + (1 - 0.0654498) * 0.01 ** (20 / 16))) < 2e-3
