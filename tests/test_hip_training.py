"""End-to-end parity of the training loop: the HIP harness (tinynerf_amd.run.Trainer) against the CPU port of
the reference's train() (oracle/torch_port.reference_training) on identical rays in deterministic mode --
BASELINE config 1 scale (Vanilla NeRF, 64x64 views, 32 samples/ray).  Checks batch structure (bit-exact
counts), per-step loss, and PSNR at equal step count (gate: 0.1 dB, north star)."""
import numpy as np
import pytest
import torch

from oracle import torch_port as tp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene():
    from tinynerf_amd import rays
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=2, res=64, seed=3, device="cpu")
    return o.contiguous(), d.contiguous(), rgb.contiguous()


def test_vanilla_training_matches_cpu_port():
    from tinynerf_amd.run import TrainConfig, Trainer, psnr
    o, d, rgb = _scene()
    n_steps = 12
    cfg = TrainConfig(method="vanilla", scene_type="aabb", batch_size=256, n_samples=32, seed=1, occupancy_res=32,
                      deterministic=True)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.renderer.state_dict().items()}
    ref_losses, ref_sd, ref_counts = tp.reference_training(sd0, o.numpy(), d.numpy(), rgb.numpy(), method="vanilla", batch_size=256,
                                                           n_samples=32, n_steps=n_steps, occupancy_res=32)
    losses, counts = [], []
    for _ in range(n_steps):
        st = tr.step()
        losses.append(tr.loss_value())
        counts.append((int(st["n_samples"]), int(st["n_rays"])))
    assert counts[0] == ref_counts[0]                       # same dynamic batch on the first step (bit-exact sampler + rule)
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-2)          # Adam amplifies ulp-level gradient differences slowly
    # (the reference recipe's lr 1e-2 drives this 10-layer stack into the all-masked branch within a few steps --
    #  on the CPU port exactly as here; the run therefore also exercises core.py:251-254)
    # PSNR at equal step count on a held-out set of rays
    test_idx = torch.arange(0, o.size(0), 7)
    with torch.no_grad():
        img = tr.render_rays(o[test_idx].to(DEV), d[test_idx].to(DEV), batch_size=512).cpu()
    p_hip = float(psnr(img, rgb[test_idx]))
    from oracle import tinynerf_oracle as orc
    # render the same rays with the CPU-trained parameters through the CPU port
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    grid = tr.occupancy_grid.grid.cpu().numpy()
    packed, info = orc.ray_provider(o[test_idx].numpy(), d[test_idx].numpy(), marcher="aabb", contraction="aabb", grid=grid,
                                    threshold=tr.occupancy_grid.threshold, n_samples=32, near=0.1, aabb=aabb)
    with torch.no_grad():
        img_ref = tp.render(ref_sd, torch.from_numpy(packed), torch.from_numpy(info), torch.ones(3), vanilla_freqs=10)
    p_ref = float(psnr(img_ref, rgb[test_idx]))
    assert abs(p_hip - p_ref) < 0.1, (p_hip, p_ref)


def test_kplanes_training_matches_cpu_port(heads):
    """K-Planes recipe (fused render path, TV regulariser, plane-gradient scatter, Adam on channel-last planes)
    with 16/32/64 planes so the CPU port finishes in seconds."""
    from tinynerf_amd.run import TrainConfig, Trainer, psnr
    o, d, rgb = _scene()
    n_steps = 24
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=2, occupancy_res=32,
                      deterministic=True, kplanes_resolutions=(16, 32, 64))
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    sd0 = {k: v.detach().cpu().contiguous().clone() for k, v in tr.renderer.state_dict().items()}
    probe = torch.arange(0, o.size(0), 5)
    with torch.no_grad():
        p_init = float(psnr(tr.render_rays(o[probe].to(DEV), d[probe].to(DEV), batch_size=1024).cpu(), rgb[probe]))
    ref_losses, ref_sd, ref_counts = tp.reference_training(sd0, o.numpy(), d.numpy(), rgb.numpy(), method="kplanes", batch_size=256,
                                                           n_samples=32, n_steps=n_steps, occupancy_res=32)
    losses, counts = [], []
    for _ in range(n_steps):
        st = tr.step()
        losses.append(tr.loss_value())
        counts.append((int(st["n_samples"]), int(st["n_rays"])))
    assert counts[0] == ref_counts[0]
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(losses, ref_losses, rtol=3e-2)      # consecutive (unshuffled) rays: the loss follows the image rows
    test_idx = torch.arange(0, o.size(0), 5)
    with torch.no_grad():
        img = tr.render_rays(o[test_idx].to(DEV), d[test_idx].to(DEV), batch_size=1024).cpu()
    from oracle import tinynerf_oracle as orc
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    packed, info = orc.ray_provider(o[test_idx].numpy(), d[test_idx].numpy(), marcher="aabb", contraction="aabb",
                                    grid=tr.occupancy_grid.grid.cpu().numpy(), threshold=tr.occupancy_grid.threshold,
                                    n_samples=32, near=0.1, aabb=aabb)
    with torch.no_grad():
        img_ref = tp.render(ref_sd, torch.from_numpy(packed), torch.from_numpy(info), torch.ones(3))
    p_hip, p_ref = float(psnr(img, rgb[test_idx])), float(psnr(img_ref, rgb[test_idx]))
    assert abs(p_hip - p_ref) < 0.1, (p_hip, p_ref)          # north-star gate: PSNR@step within 0.1 dB
    for _ in range(100):                                      # and it learns: keep training on the GPU only
        tr.step()
    with torch.no_grad():
        p_late = float(psnr(tr.render_rays(o[probe].to(DEV), d[probe].to(DEV), batch_size=1024).cpu(), rgb[probe]))
    assert p_late > p_init + 3.0, (p_init, p_late)


def test_kplanes_full_resolution_training_matches_cpu_port():
    """BASELINE config 3's model as trained (128 / 256 / 512 planes, 33 M parameters, fused gather / scatter launches, TV folded
    into Adam on channel-last planes) against the CPU port of train(): per-step loss, batch structure, and PSNR at equal step
    count within 0.1 dB (north star)."""
    from tinynerf_amd.run import TrainConfig, Trainer, psnr
    from oracle import tinynerf_oracle as orc
    o, d, rgb = _scene()
    n_steps = 8
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=6, occupancy_res=32, deterministic=True)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    assert tr.renderer.feature_module.planes[2][0].plane.shape == (1, 32, 512, 512)
    sd0 = {k: v.detach().cpu().contiguous().clone() for k, v in tr.renderer.state_dict().items()}
    ref_losses, ref_sd, ref_counts = tp.reference_training(sd0, o.numpy(), d.numpy(), rgb.numpy(), method="kplanes", batch_size=256,
                                                           n_samples=32, n_steps=n_steps, occupancy_res=32)
    losses, counts = [], []
    for _ in range(n_steps):
        st = tr.step()
        losses.append(tr.loss_value())
        counts.append((int(st["n_samples"]), int(st["n_rays"])))
    assert counts[0] == ref_counts[0]
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(losses, ref_losses, rtol=3e-2)
    test_idx = torch.arange(0, o.size(0), 5)
    with torch.no_grad():
        img = tr.render_rays(o[test_idx].to(DEV), d[test_idx].to(DEV), batch_size=1024).cpu()
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    packed, info = orc.ray_provider(o[test_idx].numpy(), d[test_idx].numpy(), marcher="aabb", contraction="aabb",
                                    grid=tr.occupancy_grid.grid.cpu().numpy(), threshold=tr.occupancy_grid.threshold,
                                    n_samples=32, near=0.1, aabb=aabb)
    with torch.no_grad():
        img_ref = tp.render(ref_sd, torch.from_numpy(packed), torch.from_numpy(info), torch.ones(3))
    p_hip, p_ref = float(psnr(img, rgb[test_idx])), float(psnr(img_ref, rgb[test_idx]))
    assert abs(p_hip - p_ref) < 0.1, (p_hip, p_ref)


def test_kplanes_full_resolution_learns_with_random_batches():
    """The production path end to end at BASELINE config 3's model size: shuffled ray draws, sampling jitter from the device RNG,
    occupancy refreshes, fused gather / scatter launches, TV in the Adam pass -- 400 steps must lift the held-out PSNR by > 6 dB
    and leave every parameter finite."""
    from tinynerf_amd import rays
    from tinynerf_amd.run import TrainConfig, Trainer, psnr
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=6, res=96, seed=11, device=DEV)
    hold = torch.arange(0, o.size(0), 9, device=DEV)
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=512, n_samples=128, seed=3, occupancy_res=64)
    tr = Trainer(cfg, o, d, rgb, torch.ones(3, device=DEV), torch.device(DEV))
    tr.occupancy_grid_updates = 20                         # 20 refreshes: empty cells decay below the threshold after 16 (run.py:104)
    with torch.no_grad():
        p0 = float(psnr(tr.render_rays(o[hold], d[hold], batch_size=1024), rgb[hold]))
    for _ in range(400):
        tr.step()
    with torch.no_grad():
        p1 = float(psnr(tr.render_rays(o[hold], d[hold], batch_size=1024), rgb[hold]))
    assert all(bool(torch.isfinite(p).all()) for p in tr.renderer.parameters())
    assert np.isfinite(tr.loss_value()) and p1 > p0 + 6.0, (p0, p1)
    assert 0.0 < tr.occupancy_grid.occupancy() < 1.0                      # the refreshes carved the grid


def test_train_entry_point_on_a_scene_on_disk(tmp_path):
    """train() end to end on a Blender-format scene written to disk: loader -> device ray tables -> training
    loop -> test render -> metrics_*.json + model.pt; the checkpoint loads back with reference key names."""
    import json
    from PIL import Image
    from tinynerf_amd import data, rays
    from tinynerf_amd.run import TrainConfig, train
    o, d, rgb, K, cams = rays.synthetic_scene(n_views=3, res=48, seed=5, device="cpu")
    imgs = (rgb.reshape(3, 48, 48, 3) * 255).to(torch.uint8).numpy()
    (tmp_path / "train").mkdir()
    frames = []
    for i in range(3):
        Image.fromarray(imgs[i]).save(tmp_path / "train" / f"r_{i}.png")
        frames.append({"file_path": f"./train/r_{i}", "transform_matrix": cams[i].tolist()})
    for split in ("train", "test"):
        json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames[:3 if split == "train" else 1]},
                  open(tmp_path / f"transforms_{split}.json", "w"))
    dev = torch.device(DEV)
    train_rays = data.RaysDataset(data.parse_nerf_synthetic(tmp_path, "train"), dev)
    test_set = data.PoseDataset(data.parse_nerf_synthetic(tmp_path, "test"), dev)
    out = tmp_path / "out"; out.mkdir()
    cfg = TrainConfig(method="kplanes", batch_size=512, n_samples=64, occupancy_res=32, kplanes_resolutions=(16, 32, 64), seed=3)
    tr, tm, em, testm = train(cfg, train_rays, None, test_set, out, max_steps=150, log_every=50)
    assert (out / "model.pt").exists() and (out / "metrics_train.json").exists() and (out / "metrics_test.json").exists()
    assert (out / "test_full_0000.png").exists()
    logged = json.load(open(out / "metrics_train.json"))
    assert len(logged) == 151 and set(logged[0]) == {"loss", "occupancy"}          # one {loss, occupancy} per step (run.py:262-266)
    assert logged[-1]["loss"] < logged[0]["loss"] and 0.0 < logged[-1]["occupancy"] <= 1.0
    assert testm[0]["psnr"] > 18.0, testm                      # the synthetic ball is learnt in 150 steps
    sd = torch.load(out / "model.pt")
    assert "feature_module.planes.0.0.plane" in sd and "rgb_decoder.net.net.5.weight" in sd
    tr.renderer.load_state_dict(sd)


@pytest.mark.parametrize("method", ["vanilla", "kplanes", "cobafa"])
def test_inference_chunking_and_no_training_state(method, heads):
    """infer(): rays are independent, so the chunk size must not change one output bit (the default walks an image in 2^16-ray
    chunks instead of the reference's training batch size, run.py:35-43); and inside torch.no_grad() no module may take its
    training forward (the wide stacks' activation workspace is 10 KB per sample)."""
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    dev = torch.device(DEV)
    cfg = TrainConfig(method=method, scene_type="aabb", batch_size=128, n_samples=64, seed=5, occupancy_res=32)
    tr = Trainer(cfg, o.to(dev), d.to(dev), rgb.to(dev), torch.ones(3, device=dev), dev)
    for _ in range(2):
        tr.step()
    oo, dd = o[:6000].to(dev), d[:6000].to(dev)
    a = tr.render_rays(oo, dd, batch_size=100)                       # ragged last chunk
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    b = tr.render_rays(oo, dd)                                        # default: one chunk here
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    n_kept = int(tr.ray_provider(oo, dd, training=False)[0].size(0))
    assert n_kept > 10000
    # packed samples, features, head outputs: under 3 KB per kept sample (+ 16 MiB of per-ray scratch), plus -- Vanilla -- the two
    # ping-pong row buffers of the layer-kernel inference forward (tn_mlp_fwd_ws: 2.3 KB per sample, at most 2^22 samples at a
    # time); the training workspace of the Vanilla stack alone would be 11.5 KB per sample
    peak = torch.cuda.max_memory_allocated() - base
    per_sample = 3072 + (2432 if method == "vanilla" else 0)
    assert peak < n_kept * per_sample + (16 << 20), (peak / n_kept, n_kept)


def test_random_ray_stream_walks_shuffled_epochs():
    """run.py:116-122: DataLoader(shuffle=True) -- every ray exactly once per epoch, in loader batches of B; the device-side
    stream (Trainer._epoch_block) must consume its permutation without gaps or repeats across steps, block redraws and the
    epoch boundary."""
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    dev = torch.device(DEV)
    n = o.size(0)
    tag = torch.zeros(n, 3)
    tag[:, 0] = torch.arange(n, dtype=torch.float32)                  # the target colour carries the ray index
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=128, n_samples=32, seed=9, occupancy_res=32, kplanes_resolutions=(16, 32, 64))
    tr = Trainer(cfg, o.to(dev), d.to(dev), tag.to(dev), torch.ones(3, device=dev), dev)
    seen = []
    while sum(len(s) for s in seen) < 2 * n + 1000:
        packed, info, target, k = tr.build_batch()
        assert info.size(0) == k * cfg.batch_size
        seen.append(target[:, 0].long().cpu())
    seen = torch.cat(seen)
    for e in range(2):                                                # two full epochs: each a permutation of all rays
        assert torch.equal(torch.sort(seen[e * n:(e + 1) * n]).values, torch.arange(n))
    assert not torch.equal(seen[:n], seen[n:2 * n])                   # reshuffled


def test_gate_ring_wraps_and_an_all_masked_step_touches_nothing():
    """The harness' device-side "Empty iteration" (core.py:251-254 + torch.optim.Adam skipping grad-is-None parameters): the flag lives
    in a ring slot raised by the weights kernel; across the ring's wrap-around the flag is 1 for ordinary steps, and a step whose
    samples are ALL masked (termination threshold above 1) leaves the decoders' parameters, Adam moments and step counts bit for
    bit where they were (the planes still take the regulariser's step, as in the reference), and the next ordinary step moves
    them again."""
    import functools
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=2, occupancy_res=32, deterministic=True)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    tr._gate_tick = tr._gate_ring.numel() - 2                     # two steps before the wrap
    for _ in range(4):
        tr.step()
        st = tr.renderer._stats
        assert st["pre_gated"] and float(st["gate"].item()) == 1.0
    assert tr._gate_tick == tr._gate_ring.numel() + 2 and float(tr._gate_ring.sum().item()) == 2.0    # zeroed at the wrap, two slots raised since

    heads = [(k, p) for k, p in tr.renderer.named_parameters() if not k.startswith("feature_module.")]
    planes = [(k, p) for k, p in tr.renderer.named_parameters() if k.startswith("feature_module.")]

    def snapshot(named):
        tr.optimizer.sync_step_counts()
        out = {}
        for k, p in named:
            st = tr.optimizer.state[p]
            out[k] = (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"]))
        return out

    before, planes_before = snapshot(heads), snapshot(planes)
    fwd = tr.renderer.forward
    tr.renderer.forward = functools.partial(fwd, early_termination_threshold=2.0)      # T = 1 is not > 2: every weight is 0
    tr.step()
    tr.renderer.forward = fwd
    assert float(tr.renderer._stats["gate"].item()) == 0.0
    after, planes_after = snapshot(heads), snapshot(planes)
    for k in before:                 # the decoders: param.grad is None in the reference -> torch.optim.Adam skips them
        assert all(torch.equal(a, b) for a, b in zip(before[k][:3], after[k][:3])) and before[k][3] == after[k][3], k
    for k in planes_before:          # the planes still receive the regulariser's gradient (run.py:254-256) and take their step
        assert not torch.equal(planes_before[k][0], planes_after[k][0]) and planes_after[k][3] == planes_before[k][3] + 1, k
    tr.step()
    moved = snapshot(heads)
    assert float(tr.renderer._stats["gate"].item()) == 1.0
    assert any(not torch.equal(after[k][0], moved[k][0]) for k in after)
    assert all(moved[k][3] == after[k][3] + 1 for k in after)


def test_gate_slot_is_fresh_for_every_forward_and_a_foreign_loss_is_gated_by_the_node():
    """Round-3 advice: (i) a second forward on the SAME trainer-built batch (another termination threshold) must not inherit the
    first forward's raised flag; (ii) a loss other than the trainer's own (no ``tn_mse_grad_gated``) on a trainer-built batch in an
    all-masked step still yields the reference's "Empty iteration" -- zero gradients everywhere (core.py:251-254) -- because the
    render node gates by itself unless the caller declared its gradient gated."""
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=2, occupancy_res=32, deterministic=True,
                      kplanes_resolutions=(16, 32, 64))
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    tr.renderer.train()
    packed, info, target, _ = tr.build_batch()
    out = tr.renderer(packed, info)
    assert float(tr.renderer._stats["gate"].item()) == 1.0
    out2 = tr.renderer(packed, info, early_termination_threshold=2.0)            # every weight 0: the flag must read 0 again
    assert float(tr.renderer._stats["gate"].item()) == 0.0
    for p in tr.renderer.parameters():
        p.grad.zero_()
    torch.nn.functional.mse_loss(out2, target).backward()                        # a foreign, ungated loss
    assert all(float(p.grad.abs().max()) == 0.0 for p in tr.renderer.parameters())
    out3 = tr.renderer(packed, info)                                             # and an ordinary forward raises it again
    assert float(tr.renderer._stats["gate"].item()) == 1.0
    torch.nn.functional.mse_loss(out3, target).backward()
    assert any(float(p.grad.abs().max()) > 0.0 for p in tr.renderer.parameters())
    assert torch.equal(out, out3)
    del out


def test_loss_accumulator_ring_survives_an_external_train_step():
    """Round-3 advice: the loss / regulariser accumulator row is picked by a counter of its own -- assigning ``train_step`` from
    outside (a resume, tests/test_hip_multigpu.py) must not land on a row that still holds an earlier step's sums."""
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=4, occupancy_res=32, deterministic=True,
                      kplanes_resolutions=(16, 32, 64))
    losses = []
    for jump in (False, True):
        tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
        tr.step(); tr.step(); tr.step()
        if jump:
            tr.train_step = 1                      # rows 1, 2 of a train_step-indexed ring would be hit a second time
            tr.occupancy_grid_updates = 10 ** 9    # (no second refresh: the parameters must follow the same trajectory)
        else:
            tr.occupancy_grid_updates = 10 ** 9
        tr.step()
        losses.append(tr.loss_value())
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[0]), losses      # (atomics order: not bit-equal; a stale row would double it)


def test_trainer_batch_with_device_rng_matches_the_oracle_bit_for_bit():
    """The dynamic batch exactly as training builds it (shuffled device-side ray draw, sampling jitter from the device's counter RNG,
    exit shortcut and coarse reject on, rule + scan in one launch, pack behind the read-back) against the oracle's restatement of
    run.py:215-244 over core.py:165-188 with the restated RNG (``orc.sampler_jitter``): same k, same ``packing_info``, same packed
    bits, same target colours.  The target colour carries the ray index, so the oracle sees the rays the trainer drew."""
    from tinynerf_amd.run import TrainConfig, Trainer
    from oracle import tinynerf_oracle as orc
    o, d, rgb = _scene()
    dev = torch.device(DEV)
    n = o.size(0)
    tag = torch.zeros(n, 3)
    tag[:, 0] = torch.arange(n, dtype=torch.float32)
    B, S = 128, 64
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=B, n_samples=S, seed=13, occupancy_res=32, kplanes_resolutions=(16, 32, 64))
    tr = Trainer(cfg, o.to(dev), d.to(dev), tag.to(dev), torch.ones(3, device=dev), dev)
    lin = torch.linspace(-1, 1, 32)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    grid = torch.where(xx * xx + yy * yy + zz * zz < 0.3, 1.0, 0.003) * (0.5 + 0.5 * torch.rand(32, 32, 32, generator=torch.Generator().manual_seed(1)))
    tr.occupancy_grid.grid.copy_(grid.to(dev))
    tr.occupancy_grid.mean = float(tr.occupancy_grid.grid.mean().item())
    aabb = np.array([[-1.5] * 3, [1.5] * 3], np.float32)
    o_np, d_np = o.numpy(), d.numpy()
    for it in range(3):
        packed, info, target, k = tr.build_batch()
        # the seed of the pass that produced this batch: a pure function of (cfg.seed, batch number, rank), whether or not the first
        # candidate block was too small and had to be redrawn (it is, here: k = 13 > the initial guess)
        from tinynerf_amd.run import jitter_seed
        seed = jitter_seed(cfg.seed, it, 0)
        assert tr.last_plan_seed == seed
        idx = target[:, 0].long().cpu().numpy()
        R = idx.shape[0]
        assert R == k * B
        jit = orc.sampler_jitter(seed, R, S)                                  # counter = (ray in the block) * S + candidate
        prov_calls = []

        def prov(oo, dd):
            r0 = len(prov_calls) * B
            prov_calls.append(r0)
            return orc.ray_provider(oo, dd, marcher="aabb", contraction="aabb", grid=tr.occupancy_grid.grid.cpu().numpy(),
                                    threshold=float(tr.occupancy_grid.threshold), n_samples=S, near=0.1, aabb=aabb, jitter=jit[r0:r0 + B])
        batches = ((o_np[idx[b:b + B]], d_np[idx[b:b + B]], tag.numpy()[idx[b:b + B]]) for b in range(0, R, B))
        ref_packed, ref_info, ref_target, ref_k = orc.dynamic_batch(batches, prov, B * S)
        assert ref_k == k, (it, ref_k, k)
        bad = np.nonzero((info.cpu().numpy() != ref_info).any(1))[0]
        assert bad.size == 0, (it, bad[:8], info.cpu().numpy()[bad[:4]], ref_info[bad[:4]], seed)
        assert np.array_equal(packed.cpu().numpy().view(np.int32), ref_packed.view(np.int32))
        assert np.array_equal(target.cpu().numpy(), ref_target)
        tr._pending = None                                                    # (no prefetch in this test: every block is drawn here)


@pytest.mark.parametrize("method,res,n_steps", [("kplanes", (16, 32, 64), 10), ("vanilla", None, 6), ("cobafa", None, 6)])
def test_gradients_along_the_reference_trajectory(method, res, n_steps, matmul):
    """Round-3 verdict: the free-running trajectory tests above hold steps >= 1 to 3e-2 because Adam amplifies ulp-level gradient
    differences.  Here the REFERENCE update is applied on both sides: at every step the HIP trainer is loaded with the CPU port's
    current parameters and given the CPU port's dynamic batch, and only the step's gradient (of the 2^10-scaled image loss,
    run.py:259) and its loss are compared -- 1e-4 of each tensor's largest element up to ReLU tie units and the reference's own
    weights-backward conditioning (capped), at EVERY step of the trajectory, for all three model configurations and both matrix
    modes.  The regulariser's gradient is folded into the Adam pass on the HIP side (tests/test_hip_models.py pins it there)."""
    from _ties import assert_grads_match_up_to_relu_ties
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    kw = dict(kplanes_resolutions=res) if res else {}
    cfg = TrainConfig(method=method, scene_type="aabb", batch_size=256, n_samples=32, seed=5, occupancy_res=32, deterministic=True, **kw)
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    if method == "cobafa":
        tr.renderer.feature_module.dropout.p = 0.0              # (the process RNG cannot be shared with the port)
    sd0 = {k: v.detach().cpu().contiguous().clone() for k, v in tr.renderer.state_dict().items()}
    cf = tr.renderer.feature_module.freqs if method == "cobafa" else None
    vf = 10 if method == "vanilla" else 0
    bg = torch.ones(3)
    got, hip_losses, checked = {}, [], []
    tr.grad_hook = lambda t: got.update({k: p.grad.detach().cpu().numpy().copy() for k, p in t.renderer.named_parameters()})

    def on_step(step, sd, packed, info, target):
        cur = {k: v.detach().clone() for k, v in sd.items()}
        tr.renderer.load_state_dict(cur)
        tr.step_on_batch(torch.from_numpy(packed).to(DEV), torch.from_numpy(info).to(DEV), torch.from_numpy(target).to(DEV), prefetch=False)
        hip_losses.append(tr.loss_value())
        pk, inf_, tg = torch.from_numpy(packed), torch.from_numpy(info), torch.from_numpy(target)

        def ref():
            return tp.grads_of(cur, lambda p: cfg.grad_scale * torch.nn.functional.mse_loss(
                tp.render(p, pk, inf_, bg, vanilla_freqs=vf, cobafa_freqs=cf), tg))[0]
        if not ref():
            # "Empty iteration" (core.py:251-254: every sample masked -- the recipe's lr drives the 10-layer stack there within a
            # few steps, on the CPU port exactly as here): the reference leaves every param.grad at None; here every gradient is 0
            assert all(float(np.abs(v).max()) == 0.0 for v in got.values()), step
            checked.append((step, -1))
            return
        flips = assert_grads_match_up_to_relu_ties(dict(got), ref, 1e-4, weights_conditioning=True, cond_cap=2e-3)
        checked.append((step, flips))
    ref_losses, _, _ = tp.reference_training(sd0, o.numpy(), d.numpy(), rgb.numpy(), method=method, batch_size=256, n_samples=32,
                                             n_steps=n_steps, occupancy_res=32, cobafa_freqs=cf, on_step=on_step,
                                             lr=1e-2 if method == "kplanes" else 1e-3)
    # (the recipe's lr 1e-2 drives the two deep stacks into the all-masked branch after two steps on this scene -- in the port as
    # here; with 1e-3 the trajectory keeps real gradients.  The HIP optimizer plays no part: parameters come from the port.)
    assert len(checked) == n_steps and sum(1 for _, f in checked if f >= 0) >= 4, checked       # steps with a real gradient
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=2e-5)     # every step's loss (MSE + TV), on identical parameters


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["vanilla", "cobafa"])
def test_heads_read_the_stack_output_from_workspace_rows(method, monkeypatch):
    """Round 4: behind a 256- / 128-wide stack the f16x2 heads take their first-layer operands from the stack's workspace rows
    (tn_mlp_desc::x_rows in tn_mlp_fwd_stash) and, from the second forward on, the stack no longer writes the row-major copy
    (TN_MLP_ROWS_ONLY + TN_MLP_X_FROM_ROWS).  Same values through the same arithmetic: the loss of every step equals the
    row-major run's (TN_ROWS_HANDOFF=0) up to the order of the gradient atomics behind the earlier updates; the row-major
    tensor really stays unwritten (poisoned with NaN here: nothing may read it)."""
    from tinynerf_amd import fused, models
    from tinynerf_amd.run import TrainConfig, Trainer
    if models.MATMUL != "f16x2":
        pytest.skip("row handoff is the f16x2 heads' path")
    o, d, rgb = _scene()
    losses = {}
    for handoff in (False, True):
        monkeypatch.setattr(fused, "ROWS_HANDOFF", handoff)
        cfg = TrainConfig(method=method, scene_type="aabb", batch_size=256, n_samples=32, seed=3, occupancy_res=32, deterministic=True)
        tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
        for group in tr.optimizer.param_groups:          # (the recipe's 1e-2 masks every sample of this scene within two steps)
            group["lr"] = 1e-3
        if method == "cobafa":
            tr.renderer.feature_module.dropout.p = 0.0
        seen = []
        if handoff:
            real = models._empty_rows

            def poisoned(n, cols, dev):                    # the stack's y comes from here: reading it would spread NaN
                t = real(n, cols, dev)
                if cols in (128, 256):
                    t.fill_(float("nan"))
                return t
            monkeypatch.setattr(models, "_empty_rows", poisoned)
        ls = []
        for _ in range(4):
            tr.step()
            ls.append(tr.loss_value())
            prod = tr.renderer.__dict__.get("_rows_producer")
            seen.append(bool(prod is not None and prod.__dict__["scratch"][2].get("rows_only")))
        losses[handoff] = ls
        if handoff:
            assert seen == [False, True, True, True], seen          # the first forward discovers the producer
            monkeypatch.setattr(models, "_empty_rows", real)
        else:
            assert not any(seen)
    assert all(np.isfinite(losses[True])), losses
    np.testing.assert_allclose(losses[True][0], losses[False][0], rtol=1e-6)
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["vanilla", "cobafa"])
def test_paired_data_gradient_behind_the_wide_stacks(method, monkeypatch):
    """Round 5: behind a wide stack the heads' backward passes go through ONE call (tn_mlp_bwd_pair with row views): the first layers'
    x-column weight gradients of both heads in one launch (x rows read once), and behind the 128-wide stack both data gradients in one pass
    (d loss / d feat written once).  Same products in the same arithmetic: parameters after three steps equal the two-call form's to
    fp32 rounding, and the first step's loss is identical."""
    from tinynerf_amd import fused
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    out = {}
    for mode in (False, True):
        monkeypatch.setattr(fused, "HEADS_PAIR_BACKWARD", mode)
        cfg = TrainConfig(method=method, scene_type="aabb", batch_size=256, n_samples=32, seed=3, occupancy_res=32, deterministic=True)
        tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
        for group in tr.optimizer.param_groups:
            group["lr"] = 1e-3
        if method == "cobafa":
            tr.renderer.feature_module.dropout.p = 0.0
        ls = []
        for _ in range(3):
            tr.step()
            ls.append(tr.loss_value())
        out[mode] = (ls, {k: v.detach().float().cpu().numpy().copy() for k, v in tr.renderer.state_dict().items()})
    assert out[False][0][0] == out[True][0][0]
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-5)
    for k, v in out[False][1].items():
        w = out[True][1][k]
        assert np.all(np.isfinite(w)), k
        diff = np.abs(w - v)            # (see test_last_layer_merged_into_the_heads on Adam and near-zero gradients)
        tol = 2e-5 * max(1e-3, float(np.abs(v).max()))
        assert float((diff > tol).mean()) < 1e-3 and float(diff.max()) < 2e-4, (k, float(diff.max()), float((diff > tol).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["vanilla", "cobafa"])
def test_last_layer_merged_into_the_heads(method, monkeypatch):
    """Round 5 (TN_MLP_SKIP_LAST): the stack's last layer is Linear(F, F) and both heads begin with a Linear on its output (reference
    models.py:59-89, 239-247) -- from the second training forward on the harness folds the former into the latter (merged parameters once per
    step, the stack stops at its last hidden activation, the feature tensor never exists).  Same function, other association of the
    products: every step's loss and all parameters after four steps equal the unmerged run's to fp32 rounding, the last layer's
    parameters included (their gradients arrive through the merge by the chain rule)."""
    from tinynerf_amd import fused, models
    from tinynerf_amd.run import TrainConfig, Trainer
    if models.MATMUL != "f16x2":
        pytest.skip("the merge rides on the f16x2 heads' row handoff")
    o, d, rgb = _scene()
    out = {}
    for merge in (False, True):
        monkeypatch.setattr(fused, "MERGE_LAST", merge)
        cfg = TrainConfig(method=method, scene_type="aabb", batch_size=256, n_samples=32, seed=3, occupancy_res=32, deterministic=True)
        tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
        for group in tr.optimizer.param_groups:
            group["lr"] = 1e-3
        if method == "cobafa":
            tr.renderer.feature_module.dropout.p = 0.0
        ls, skipped = [], []
        for _ in range(4):
            tr.step()
            ls.append(tr.loss_value())
            prod = tr.renderer.__dict__.get("_rows_producer")
            skipped.append(bool(prod is not None and prod.__dict__["scratch"][2].get("skipped_last")))
        assert skipped == ([False, True, True, True] if merge else [False] * 4), skipped
        out[merge] = (ls, {k: v.detach().float().cpu().numpy().copy() for k, v in tr.renderer.state_dict().items()})
    assert out[True][0][0] == out[False][0][0]
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=5e-5)
    moved = 0
    for k, v in out[False][1].items():
        w = out[True][1][k]
        assert np.all(np.isfinite(w)), k
        # (Adam divides by sqrt(v): where a gradient is a rounding away from 0 the two associations may step in different directions --
        # a handful of elements may differ by a few lr * 1e-3; everything else agrees to fp32 rounding)
        diff = np.abs(w - v)
        tol = 5e-5 * max(1e-3, float(np.abs(v).max()))
        assert float((diff > tol).mean()) < 1e-3 and float(diff.max()) < 2e-4, (k, float(diff.max()), float((diff > tol).mean()))
        moved += 1
    assert moved > 10


@pytest.mark.gpu
def test_linear_merge_kernels_against_autograd():
    """tn_linear_merge_fwd / _bwd == the torch expressions they replace and their autograd (fp64 reference)."""
    from tinynerf_amd import fused
    torch.manual_seed(5)
    for F, pe in ((256, 51), (128, 51)):
        ps = [torch.randn(64, pe + F, device=DEV) * 0.1, torch.randn(64, device=DEV), torch.randn(64, F, device=DEV) * 0.1, torch.randn(64, device=DEV),
              torch.randn(F, F, device=DEV) * 0.1, torch.randn(F, device=DEV)]
        for p in ps:
            p.requires_grad_(True)
        outs = fused._MergeLast.apply(False, pe, *ps)
        gs = [torch.randn_like(o_) for o_ in outs]
        got = torch.autograd.grad(outs, ps, gs)
        ref_p = [p.detach().double().requires_grad_(True) for p in ps]
        wc, bc, ws, bs, wl, bl = ref_p
        ref = (torch.cat([wc[:, :pe], wc[:, pe:] @ wl], 1), bc + wc[:, pe:] @ bl, ws @ wl, bs + ws @ bl)
        want = torch.autograd.grad(ref, ref_p, [g.double() for g in gs])
        for a_, b_ in zip(outs, ref):
            np.testing.assert_allclose(a_.detach().cpu().numpy(), b_.detach().cpu().numpy(), rtol=0, atol=2e-6 * float(b_.abs().max()))
        for a_, b_ in zip(got, want):
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=0, atol=2e-6 * float(b_.abs().max()))


@pytest.mark.gpu
def test_planes_optimizer_pass_beside_the_weight_gradient_kernels(monkeypatch):
    """TN_ADAM_OVERLAP (default on, N == 1): the planes' optimizer pass starts on a stream of its own once the chain + scatter launch has made
    their gradients final, beside the heads' weight-gradient kernels; the remaining parameters follow as before.  Same arithmetic on the
    same values: losses and parameters equal the serial run's to the order noise of the gradient atomics."""
    from tinynerf_amd import run
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    out = {}
    for overlap in (False, True):
        monkeypatch.setattr(run, "ADAM_OVERLAP", overlap)
        cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=3, occupancy_res=32, deterministic=True,
                          kplanes_resolutions=(32, 64))
        tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
        ls = []
        for _ in range(4):
            tr.step()
            ls.append(tr.loss_value())
        torch.cuda.synchronize()
        assert (tr._side2 is not None) == overlap
        out[overlap] = (ls, {k: v.detach().float().cpu().numpy().copy() for k, v in tr.renderer.state_dict().items()})
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-4)
    for k, v in out[False][1].items():
        diff = np.abs(out[True][1][k] - v)
        assert float((diff > 1e-4 * max(1e-3, float(np.abs(v).max()))).mean()) < 5e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["plane", "colour_weight", "sigma_bias"])
def test_a_non_finite_parameter_surfaces_as_the_reference_nan_loss(where):
    """torch.relu hands a NaN on (models.py:7-28), so in the reference ONE non-finite parameter makes the loss NaN at the next step
    (run.py:252-256: MSE + TV over every plane).  The kernels' ReLU is v_max_f32, which returns 0 for a NaN pre-activation: rendered
    colours and the MSE stay finite.  Round-4 verdict: decide.  Decision (round 5): the optimizer pass raises a device flag when it
    writes a non-finite parameter (tn_adam_multi_gated, zero_grad bit 1), a non-finite plane value poisons the regulariser sums of
    tn_adam_reg_multi, and Trainer.loss_device() reports NaN from then on -- no host sync, nothing on the kernels' hot paths."""
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb = _scene()
    cfg = TrainConfig(method="kplanes", scene_type="aabb", batch_size=256, n_samples=32, seed=1, occupancy_res=32, deterministic=True,
                      kplanes_resolutions=(16, 32, 64))
    tr = Trainer(cfg, o.to(DEV), d.to(DEV), rgb.to(DEV), torch.ones(3, device=DEV), torch.device(DEV))
    for _ in range(2):
        tr.step()
        assert np.isfinite(tr.loss_value())
    with torch.no_grad():
        if where == "plane":
            tr.renderer.feature_module.plane_tensors()[4][0, 5, 17, 9] = float("nan")
        elif where == "colour_weight":
            tr.renderer.rgb_decoder.net.net[2][0].weight[3, 5] = float("inf")
        else:
            tr.renderer.sigma_decoder.net.net[0].bias[7] = float("nan")
    tr.step()
    tr.step()                                    # (the reference: NaN at the step after the parameter went bad, and ever after)
    assert np.isnan(tr.loss_value())
    tr.step()
    assert np.isnan(tr.loss_value())
