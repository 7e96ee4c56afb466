"""Training parity ACROSS occupancy refreshes (reference run.py:248-249 -> core.py:133-145 -> the sampler of the following
steps, core.py:147-188): the chain a6 -> a7 -> a8 on a grid that the refresh has actually carved.

The recipe refreshes every 16 * 4096 / B steps, cells need 16 decays to fall below the threshold, and a randomly initialised
field is spatially almost constant (sigma varies by 2 % over the box), so the recipe's first hundreds of steps only ever see the
all-ones grid.  Here both sides -- tinynerf_amd.run.Trainer and the CPU port of train() (oracle/torch_port.reference_training)
-- start from the same DESIGNED state instead: the sigma head's output layer scaled so that alpha = 1 - exp(-sigma * step)
straddles the 0.01 threshold across the box, the grid initialised to 0.012 (one decay below the threshold, so a single refresh
carves it), and a refresh every 4 steps.  Deterministic mode (consecutive rays, no jitter, voxel-centre refresh) on both sides.
"""
import numpy as np
import pytest
import torch

from oracle import torch_port as tp
from oracle import tinynerf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"

PERIOD, N_STEPS, GRID0 = 4, 10, 0.012


def _design(sd0, method, step_size, cobafa_freqs, res=32):
    """scale the sigma head's output layer by `k` and move its bias so that quantile `q` of the voxel centres sits exactly at
    the occupancy threshold (alpha = 0.01): a fraction 1 - q of the cells passes the first refresh, the rest decays"""
    sd = {k: v.clone() for k, v in sd0.items()}
    k, q = (40.0, 0.75) if method == "kplanes" else (8.0, 0.6)
    if method == "cobafa":
        sd["feature_module.coef_grid.grid"] = (sd["feature_module.coef_grid.grid"] * 1000.0).contiguous(memory_format=torch.channels_last_3d)
    sd["sigma_decoder.net.net.2.weight"] = sd["sigma_decoder.net.net.2.weight"] * k
    jit = np.full((res, res, 3), 0.5, np.float32)
    pts = torch.from_numpy(np.concatenate([orc.occupancy_voxel_coords((res,) * 3, i, jit) for i in range(res)]))
    with torch.no_grad():
        y = tp.mlp(sd, "sigma_decoder.net.net.", tp.features(sd, pts, 0, cobafa_freqs)).ravel()
    y_thr = 1.0 + float(np.log(-np.log(1.0 - 0.01) / step_size))          # alpha(y_thr) = 0.01
    sd["sigma_decoder.net.net.2.bias"] = sd["sigma_decoder.net.net.2.bias"] + (y_thr - float(torch.quantile(y, q)))
    return sd


def _run(method):
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer
    if method == "kplanes":
        o, d, rgb, _, _ = rays.synthetic_scene(n_views=2, res=64, seed=3, device="cpu")
        cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=2, occupancy_res=32,
                          deterministic=True, kplanes_resolutions=(16, 32, 64))
        bg, kw = torch.ones(3), dict(method="kplanes", scene_type="aabb")
    else:                                                              # BASELINE config 5's model and scene type
        o, d, rgb, _, _ = rays.synthetic_scene(n_views=2, res=48, seed=9, device="cpu")
        o = (o * 0.08).contiguous()
        cfg = TrainConfig(method="cobafa", scene_type="unbounded", batch_size=256, n_samples=32, seed=5, occupancy_res=32,
                          deterministic=True, scene_scale=1.3)
        bg, kw = None, dict(method="cobafa", scene_type="unbounded", scene_scale=1.3, bg=None)
    o, d, rgb = o.contiguous(), d.contiguous(), rgb.contiguous()
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), None if bg is None else bg.to(DEV), torch.device(DEV))
    freqs = None
    if method == "cobafa":
        tr.renderer.feature_module.dropout.p = 0.0
        freqs = tuple(tr.renderer.feature_module.freqs)
        kw["cobafa_freqs"] = freqs
    sd0 = {k: v.detach().cpu().contiguous().clone() for k, v in tr.renderer.state_dict().items()}
    sd1 = _design(sd0, method, float(tr.occupancy_grid.step_size), freqs)
    tr.renderer.load_state_dict(sd1)
    tr.occupancy_grid_updates = PERIOD
    tr.occupancy_grid.grid.fill_(GRID0)
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    grids_ref = []
    ref_losses, _, ref_counts = tp.reference_training(
        {k: v.contiguous() for k, v in sd1.items()}, o.numpy(), d.numpy(), rgb.numpy(), batch_size=256, n_samples=32, n_steps=N_STEPS,
        occupancy_res=32, occ_updates=PERIOD, grid0=np.full((32,) * 3, GRID0, np.float32), grids_out=grids_ref, **kw)
    losses, counts, grids = [], [], []
    for step in range(N_STEPS):
        st = tr.step()
        losses.append(tr.loss_value())
        counts.append((int(st["n_samples"]), int(st["n_rays"])))
        if step % PERIOD == 0:
            grids.append((step, tr.occupancy_grid.grid.cpu().numpy().copy(), float(tr.occupancy_grid.mean)))
    return losses, counts, grids, ref_losses, ref_counts, grids_ref


@pytest.mark.timeout(900)
@pytest.mark.parametrize("method", ["kplanes", "cobafa"])
def test_training_parity_across_occupancy_refreshes(method):
    losses, counts, grids, ref_losses, ref_counts, grids_ref = _run(method)
    assert [g[0] for g in grids] == [g[0] for g in grids_ref] == [0, 4, 8]                # three refreshes inside the run
    flips_total = 0
    for (step, g, mean), (_, gr, mean_r) in zip(grids, grids_ref):
        thr, thr_r = min(0.01, mean), min(0.01, mean_r)
        occ, occ_r = g > thr, gr > thr_r
        frac = float(occ_r.mean())
        # the first refresh carves the designed state (the optimizer may fill the grid again later: Cobafa's density grows
        # everywhere within four steps of the recipe's lr 1e-2 -- the later refreshes then pin decay chains and all-ones cells)
        assert frac > 0.15 and (step > 0 or frac < 0.85), (step, frac)
        flips = int((occ != occ_r).sum())
        flips_total += flips
        # G5's bound: cells whose alpha sits within rounding of the threshold may fall on either side (expf / MFMA summation
        # order; after a few optimizer steps the parameters themselves differ in the last digits): <= 0.2 % of the cells
        assert flips <= 0.002 * g.size, (step, flips)
        # away from flipped cells the float values are the same decay chain: 1 or GRID0 * decay^k
        same = occ == occ_r
        np.testing.assert_allclose(g[same], gr[same], rtol=1e-6, atol=0)
        assert abs(mean - mean_r) <= 2e-3 * max(mean_r, 1e-6) + 1e-6
    # the dynamic batches of the steps behind each refresh are built on the carved grid (sample_mask -> batch_plan -> scan ->
    # pack): counts equal the port's exactly when no cell flipped, and within the flipped cells' share of samples otherwise
    assert counts[0] == ref_counts[0]                                                      # step 0: built on the initial grid
    for s in range(1, N_STEPS):
        (n, r), (n_r, r_r) = counts[s], ref_counts[s]
        if flips_total == 0:
            assert (n, r) == (n_r, r_r), (s, counts, ref_counts)
        else:
            assert abs(r - r_r) <= 256 and abs(n - n_r) <= 0.01 * n_r + 64, (s, counts, ref_counts)
    if method == "kplanes":          # (Cobafa's salt-and-pepper occupancy keeps nearly every candidate: k stays 1 there)
        assert len({c[1] for c in counts[1:]}) > 1 or counts[1][1] > 256                   # dynamic batching kicked in (k > 1)
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(losses, ref_losses, rtol=5e-2)
