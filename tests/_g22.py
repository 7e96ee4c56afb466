"""Shared helpers of the G22 tests: the trace of the reference's own ``train()`` (oracle/make_train_trace.py) and the ray table
it walked.  Test-side only."""
import os

import numpy as np

from conftest import GOLDEN, load_golden


def trace(name):
    return load_golden(f"G22_reference_train_{name}")


def ray_table():
    """rays_o, rays_d, rgbs [80000, 3] fp32 and the background colour exactly as the reference's RaysDataset held them"""
    g = load_golden("G22_rays_hotdog")
    per = int(g["rays_per_image"])
    o = np.repeat(g["rays_o_per_image"], per, axis=0).astype(np.float32)
    rgbs = (g["rgbs_u8"].astype(np.float32) / np.float32(255.)).astype(np.float32)
    return o, g["rays_d"].astype(np.float32), rgbs, g["bg_color"].astype(np.float32)


def decay_ladder(decay=0.01 ** (1 / 16), n=256):
    """fp32 values a cell takes after k decays since its last reset (core.py:140-144): 1, fl(d * 1), fl(d * fl(d * 1)), ..."""
    out = [np.float32(1.0)]
    for _ in range(n - 1):
        out.append(np.float32(np.float32(decay) * out[-1]))
    return np.array(out, np.float32)


def subsample_index(numel):
    """oracle/make_train_trace.py subsample_index"""
    stride = max(1, numel // 8192)
    return np.arange(0, numel, stride, dtype=np.int64)[:8192]


def compare_final_state(g, sd, rel, what):
    """``sd``: name -> contiguous numpy array in the reference's logical layout.  Every recorded tensor to ``rel`` of its largest
    recorded magnitude; large tensors through their recorded subsample and their fp64 sums."""
    worst = {}
    for n in g["param_names"]:
        n = str(n)
        a = np.ascontiguousarray(sd[n])
        assert tuple(a.shape) == tuple(g["shape/" + n]), n
        if "final/" + n in g:
            ref = g["final/" + n]
            got = a
        else:
            ref = g["final_sub/" + n]
            got = a.ravel()[subsample_index(a.size)]
            sums = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum(), (a.astype(np.float64) ** 2).sum()])
            np.testing.assert_allclose(sums, g["final_sum/" + n], rtol=max(rel, 1e-6), err_msg=f"{what}: fp64 sums of {n}")
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        worst[n] = err / max(scale, 1e-30)
        assert err <= rel * scale + 1e-12, f"{what}: {n} differs from the reference's train() by {err / scale:.2e} of its largest value (allowed {rel:.0e})"
    return worst
