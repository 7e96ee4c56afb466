"""G22: the training-loop oracle against the reference's OWN loop.

``oracle/make_train_trace.py`` ran ``src.run.train()`` of the reference (its constructors, DataLoader, dynamic-batch loop, occupancy
refresh, Adam + ChainedScheduler, GradScaler quirk) on the reference's hotdog fixture with regenerable random streams and recorded
what it did.  Here ``oracle/torch_port.reference_training`` -- the restatement every PSNR / trajectory golden (G17 - G21) and the GPU
training tests lean on -- must reproduce that record: batch structure and learning rates exactly, losses to 1e-6, the refreshed
occupancy grids cell for cell, the parameters after the last step to 1e-5.  CPU only; the HIP harness is held to the same record in
tests/test_hip_train_trace.py."""
import hashlib
import os

import numpy as np
import pytest
import torch

import _g22
from oracle import torch_port as tp


@pytest.mark.parametrize("name", ["kplanes", "vanilla", "kplanes_lr"])
def test_initial_parameters_are_the_reference_constructors(name):
    """sha256 of every tensor the reference's constructors left behind torch.manual_seed(seed) (run.py:130-152): the port's
    ``initial_state`` and the product's ``build_renderer`` (host logic: torch initialisers on the CPU) both reproduce them"""
    g = _g22.trace(name)
    method, seed = str(g["method"]), int(g["seed"])
    sd = tp.initial_state(method, seed)
    from tinynerf_amd.run import TrainConfig, build_renderer
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        renderer, _, _ = build_renderer(TrainConfig(method=method, batch_size=int(g["batch_size"]), n_samples=int(g["n_samples"]), seed=seed),
                                        torch.ones(3), torch.device("cpu"))
    prod = {k: v.detach().contiguous() for k, v in renderer.state_dict().items()}
    assert [str(n) for n in g["param_names"]] == [k for k in sd if not k.endswith("freqs")]
    for n in g["param_names"]:
        n = str(n)
        want = str(g["init_sha256/" + n])
        assert hashlib.sha256(np.ascontiguousarray(sd[n].numpy()).tobytes()).hexdigest() == want, f"oracle initial_state: {n}"
        assert hashlib.sha256(np.ascontiguousarray(prod[n].numpy()).tobytes()).hexdigest() == want, f"build_renderer: {n}"
        if "init_sub/" + n in g:
            np.testing.assert_array_equal(sd[n].numpy().ravel()[_g22.subsample_index(sd[n].numel())], g["init_sub/" + n])


def _replay(name, n_steps=None):
    g = _g22.trace(name)
    o, d, rgbs, bg = _g22.ray_table()
    assert o.shape[0] == int(g["n_rays_table"])
    method, seed, K = str(g["method"]), int(g["seed"]), int(g["steps"]) if n_steps is None else n_steps
    grids, lrs = [], []
    losses, sd, counts = tp.reference_training(tp.initial_state(method, seed), o, d, rgbs, method=method, batch_size=int(g["batch_size"]),
                                               n_samples=int(g["n_samples"]), n_steps=K, bg=tuple(bg.tolist()), replay={"seed": seed, "rank": 0},
                                               loader="dataloader", grids_out=grids, lrs_out=lrs)
    return g, K, losses, sd, counts, grids, lrs


def _check(name, loss_rel, param_rel, n_steps=None):
    g, K, losses, sd, counts, grids, lrs = _replay(name, n_steps)
    # batch structure: the same rays, the same jitter, the same grid -> the same packed samples, exactly
    np.testing.assert_array_equal([c[0] for c in counts], g["n_samples_per_step"][:K])
    np.testing.assert_array_equal([c[1] for c in counts], g["n_rays_per_step"][:K])
    np.testing.assert_array_equal(lrs, g["lr_after_step"][:K])                      # run.py:188-199
    # occupancy refreshes (core.py:133-145): every cell on the same rung of the decay ladder
    ladder = _g22.decay_ladder()
    n_ref = int((g["grid_steps"] < K).sum())
    assert [s for s, _, _ in grids] == g["grid_steps"][:n_ref].tolist()
    for i, (s, grid, mean) in enumerate(grids):
        want = ladder[g["grid_decays"][i]]
        flips = int((grid != want).sum())
        assert flips <= 1e-5 * grid.size, f"refresh at step {s}: {flips} cells differ from the reference's grid"
        assert abs(mean - g["grid_means"][i]) <= 1e-6
    np.testing.assert_allclose(losses, g["loss"][:K], rtol=loss_rel, err_msg="per-step loss (run.py:264) against the reference's train()")
    if K == int(g["steps"]):
        worst = _g22.compare_final_state(g, {k: v.numpy() for k, v in sd.items()}, param_rel, "oracle/torch_port.reference_training")
        print(name, "largest relative parameter difference", max(worst.values()))
    print(name, "largest relative loss difference", float(np.max(np.abs(np.array(losses) / g["loss"][:K] - 1))))


def test_port_replays_the_reference_train_loop_kplanes():
    _check("kplanes", 1e-6, 1e-5)


def test_port_replays_the_reference_train_loop_vanilla():
    _check("vanilla", 1e-6, 1e-5)


@pytest.mark.skipif(not os.environ.get("TN_SLOW_CPU_TESTS"), reason="~6 min of CPU: the 70-step trace with a refresh per step and an LR milestone; "
                    "the GPU suite holds the HIP harness to it, TN_SLOW_CPU_TESTS=1 the port")
def test_port_replays_the_reference_train_loop_across_a_milestone():
    _check("kplanes_lr", 1e-6, 1e-5)


def test_port_replays_the_head_of_the_milestone_trace():
    """the first steps of the B = 65536 trace (a loader batch larger than what is left of the epoch, a refresh per step): cheap
    enough for the default CPU suite"""
    _check("kplanes_lr", 1e-6, 1e-5, n_steps=3)
