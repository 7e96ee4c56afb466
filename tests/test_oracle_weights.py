"""Pin the weights restatement (oracle/weights_ref.c, reference src/cuda.cu:3-58).

The reference kernels only run on a GPU and the reference has no CPU version, so this
restatement is pinned by: (1) C twin == python-loop twin bitwise, (2) an independent
vectorised fp64 formulation, (3) hand known answers, (4) finite-difference gradients
of the forward (threshold 0) against the analytic backward."""
import numpy as np
import pytest

from oracle import tinynerf_oracle as orc


def ragged(rng, n_rays, max_len, p_empty=0.2):
    cnt = rng.integers(1, max_len + 1, n_rays).astype(np.int32)
    cnt[rng.random(n_rays) < p_empty] = 0
    start = (np.cumsum(cnt) - cnt).astype(np.int32)
    return np.stack([start, cnt], -1), int(cnt.sum())


def test_single_sample_and_empty_ray():
    info = np.array([[0, 1], [1, 0], [1, 1]], np.int32)
    s = np.array([2.0, 3.0], np.float32); d = np.array([0.5, 0.25], np.float32)
    w = orc.weights_fwd(s, d, info, 1e-4)
    np.testing.assert_allclose(w, 1 - np.exp(-s * d), rtol=2e-7)


def test_geometric_series_and_termination_point():
    n = 64
    s = np.full(n, 4.0, np.float32); d = np.full(n, 0.25, np.float32)       # alpha = e^-1
    info = np.array([[0, n]], np.int32)
    w = orc.weights_fwd(s, d, info, 1e-4)
    kstar = int(np.ceil(-np.log(1e-4)))            # first k with e^-k <= 1e-4  -> 10
    assert kstar == 10
    expect = np.exp(-np.arange(n)) * (1 - np.exp(-1.0))
    np.testing.assert_allclose(w[:kstar], expect[:kstar], rtol=1e-5)
    assert (w[kstar:] == 0).all() and w[kstar - 1] > 0
    # threshold 0 never terminates (until T underflows)
    w0 = orc.weights_fwd(s, d, info, 0.0)
    np.testing.assert_allclose(w0[:40], expect[:40], rtol=3e-5)


@pytest.mark.parametrize("seed", [0, 1])
def test_c_equals_python_loops_and_vectorised(seed):
    rng = np.random.default_rng(seed)
    info, n = ragged(rng, 57, 40)
    s = (rng.random(n) * 30).astype(np.float32); d = (rng.random(n) * 0.05 + 0.001).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    w_c = orc.weights_fwd(s, d, info, 1e-4)
    w_py = orc.weights_fwd_py(s, d, info, 1e-4)
    np.testing.assert_allclose(w_c, w_py, rtol=1e-6, atol=2.5e-7)   # 1 ulp of alpha in (1-alpha)
    assert (w_c == 0).sum() == pytest.approx((w_py == 0).sum(), abs=2)
    np.testing.assert_allclose(w_c, orc.weights_fwd_vectorised(s, d, info, 1e-4), rtol=2e-5, atol=2.5e-7)
    assert (w_c == 0).any()                                          # termination happened
    gs_c = orc.weights_bwd(s, d, info, w_c, g)
    gs_py = orc.weights_bwd_py(s, d, info, w_c, g)
    np.testing.assert_allclose(gs_c, gs_py, rtol=2e-5, atol=1e-8)


def test_backward_matches_finite_differences():
    rng = np.random.default_rng(3)
    info, n = ragged(rng, 9, 12, p_empty=0.1)
    s = (rng.random(n) * 3).astype(np.float64); d = (rng.random(n) * 0.2 + 0.05).astype(np.float64)
    g = rng.standard_normal(n)
    f = lambda sv: float((orc.weights_fwd_vectorised(sv, d, info, 0.0) * g).sum())
    num = np.zeros(n)
    for i in range(n):
        e = np.zeros(n); e[i] = 1e-6
        num[i] = (f(s + e) - f(s - e)) / 2e-6
    w = orc.weights_fwd(s.astype(np.float32), d.astype(np.float32), info, 0.0)
    ana = orc.weights_bwd(s.astype(np.float32), d.astype(np.float32), info, w, g.astype(np.float32))
    np.testing.assert_allclose(ana, num, rtol=2e-4, atol=2e-6)


def test_backward_ignores_termination():
    """cuda.cu:49-56 keeps multiplying T over terminated samples: grad = step*T*g there."""
    n = 32
    s = np.full(n, 8.0, np.float32); d = np.full(n, 0.25, np.float32)
    info = np.array([[0, n]], np.int32)
    w = orc.weights_fwd(s, d, info, 1e-4)
    k0 = int(np.argmax(w == 0))
    g = np.ones(n, np.float32)
    gs = orc.weights_bwd(s, d, info, w, g)
    T = np.exp(-2.0 * (np.arange(n) + 1))
    np.testing.assert_allclose(gs[k0:k0 + 4], 0.25 * T[k0:k0 + 4], rtol=0, atol=3e-8)   # + fp32 residue of acc
    assert (gs[k0:k0 + 4] > 0).all()


def test_composite_matches_python():
    rng = np.random.default_rng(5)
    info, n = ragged(rng, 20, 16)
    rgb = rng.random((n, 3)).astype(np.float32); w = (rng.random(n) * 0.1).astype(np.float32)
    out = orc.composite(rgb, w, info, np.array([1, 1, 1], np.float32))
    ref = np.stack([(rgb[a:a + c] * w[a:a + c, None]).sum(0) + 1 - w[a:a + c].sum() for a, c in info])
    np.testing.assert_allclose(out, ref, atol=1e-6)
