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
