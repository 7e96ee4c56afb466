"""tinynerf_amd.config: the switches that select code paths are read once, validated, and every accepted combination is run."""
import os
import re

import numpy as np
import pytest
import torch


def test_supported_combinations_round_trip_and_others_are_refused():
    from tinynerf_amd import config
    combos = list(config.supported_combinations())
    assert len(combos) == len(set(combos)) == 98          # f16x2: 2^7 settings of the A / B switches, merge_last only with rows_handoff; + bf16x3, fp32
    for c in combos:
        assert config.Config.from_env(c.as_env()) == c
    assert config.Config.from_env({}) == config.Config()   # the defaults
    for bad in ({"TN_MATMUL": "bf16x3", "TN_KP_LEAN": "0"}, {"TN_MATMUL": "fp32", "TN_ROWS_HANDOFF": "0"}, {"TN_MATMUL": "tf32"}):
        with pytest.raises(RuntimeError):
            config.Config.from_env(bad)
    # two spellings of one behaviour are one configuration
    assert config.Config.from_env({"TN_ROWS_HANDOFF": "0"}) == config.Config.from_env({"TN_ROWS_HANDOFF": "0", "TN_MERGE_LAST": "0"})


def test_no_other_module_reads_a_switch_from_the_environment():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinynerf_amd")
    allowed = {"config.py": None, "_lib.py": {"TN_LIB_PATH", "TN_TRACE"}}      # (which library file; synchronise + name every launch: no code path)
    for fn in sorted(os.listdir(root)):
        if not fn.endswith(".py"):
            continue
        names = set(re.findall(r"environ(?:\.get)?[\(\[]\s*[\"'](TN_[A-Z0-9_]+)", open(os.path.join(root, fn)).read()))
        if fn in allowed:
            assert allowed[fn] is None or names <= allowed[fn], (fn, names)
        else:
            assert not names, (fn, names)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["kplanes", "vanilla"])
def test_every_supported_combination_trains_the_same_function(method):
    """three optimizer steps of a small trainer under each accepted configuration: the loss after them equals the default configuration's
    to 1e-3 (every switch selects another evaluation order of the same arithmetic; the three matrix modes differ by fp32 rounding)"""
    from tinynerf_amd import config, rays
    from tinynerf_amd.run import TrainConfig, Trainer
    o, d, rgb, _, _ = rays.synthetic_scene(n_views=2, res=48, seed=3, device="cuda")
    before = config.Config(matmul=__import__("tinynerf_amd.models", fromlist=["MATMUL"]).MATMUL)

    def run(cfg):
        config.apply(cfg)
        tc = TrainConfig(method=method, scene_type="aabb", batch_size=128, n_samples=32, seed=4, occupancy_res=32, deterministic=True,
                         kplanes_resolutions=(16, 32, 64))
        tr = Trainer(tc, o, d, rgb, torch.ones(3, device="cuda"), torch.device("cuda"))
        for g in tr.optimizer.param_groups:
            g["lr"] = 1e-3
        for _ in range(3):
            tr.step()
        v = tr.loss_value()
        del tr
        return v
    try:
        ref = run(config.Config())
        assert np.isfinite(ref)
        # switches that cannot matter for this method are not walked again
        relevant = ("kp_lean", "infer_pair", "adam_overlap", "side_plan") if method == "kplanes" else ("rows_handoff", "merge_last", "heads_pair", "side_plan")
        seen = set()
        for c in config.supported_combinations():
            key = (c.matmul,) + tuple(getattr(c, n) for n in relevant)
            if key in seen:
                continue
            seen.add(key)
            got = run(c)
            assert abs(got - ref) <= 1e-3 * abs(ref), (c, got, ref)
        assert len(seen) >= 8
    finally:
        config.apply(config.Config(matmul=before.matmul))
