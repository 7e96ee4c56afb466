"""CPU oracle for the tinynerf ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tinynerf_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

This is a numpy (fp32) restatement of the reference's algorithm, one function per
row of SURVEY.md section 8(a).  Each function cites the reference file:line it
follows.  The oracle is *pinned*: ``tests/golden/*.npz`` hold input/output vectors
captured from the imported reference (``oracle/make_goldens.py``, run in the build
container where ``/root/reference`` exists) and ``tests/test_oracle_golden.py``
checks every function here against them.  The weights kernel (reference
``src/cuda.cu``) cannot be executed without a GPU and has no CPU implementation in
the reference, so ``weights_fwd/bwd`` (and their C twin ``oracle/weights_ref.c``)
are a restatement of ``cuda.cu:14-28,49-56`` cross-checked by an independent
vectorised formulation and hand-computed known answers.

All arithmetic is IEEE fp32 with one rounding per operation, in the operation
order of the reference's torch expressions, so integer outputs (``packing_info``)
and the sampled coordinates are bit-exact against the torch CPU kernels.
"""
from __future__ import annotations

import ctypes
import itertools
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# a1 / a3: ray marchers
# --------------------------------------------------------------------------- #
def aabb_step_size(aabb: np.ndarray, n_samples: int) -> np.float32:
    """reference core.py:68-70 -- ||hi - lo||_2 / n_samples (fp32)."""
    ext = (aabb[1] - aabb[0]).astype(f32)
    sq = ext * ext
    acc = f32(0)
    for v in sq:
        acc = f32(acc + v)
    return f32(np.sqrt(acc) / f32(n_samples))


def march_aabb(rays_o, rays_d, aabb, n_samples, near=0.0, far=1e5):
    """reference core.py:72-88.  Returns t_values [R,S], step_sizes [R,S]."""
    o = np.asarray(rays_o, f32)
    d = np.asarray(rays_d, f32)
    aabb = np.asarray(aabb, f32)
    step = aabb_step_size(aabb, n_samples)
    dist = aabb[:, None, :] - o[None]                       # core.py:78   [2,R,3]
    denom = np.where(d == 0, d + f32(1e-9), d)              # core.py:79
    inter = dist / denom[None]
    t_min = np.max(np.min(inter, axis=0), axis=1)           # core.py:80
    t_min = np.minimum(np.maximum(t_min, f32(near)), f32(far))   # core.py:81
    steps = np.arange(n_samples, dtype=f32) * step          # core.py:84
    t = (t_min[:, None] + steps[None]).astype(f32)          # core.py:85
    delta = np.full_like(t, step)                           # core.py:86
    return t, delta


def _torch_linspace(start: float, end: float, steps: int) -> np.ndarray:
    """torch.linspace itself.  ATen's CPU kernel mixes a scalar and a SIMD formula
    (base + step*lane per vector, second half counted down from ``end``), so its bits
    depend on the host's vector width; it is third-party arithmetic the reference calls
    at core.py:53, so the oracle calls the same primitive instead of restating it."""
    import torch
    return torch.linspace(float(start), float(end), int(steps), dtype=torch.float32).numpy().copy()


def unbounded_table(n_samples, near=0.0, uniform_range=1.0):
    """reference core.py:52-55: the per-ray-identical t/step table, length S each."""
    u = _torch_linspace(0.0, 1.0 - (1.0 / (n_samples + 2)), n_samples + 1)
    with np.errstate(divide="ignore"):
        fu = np.where(u < f32(0.5), f32(2) * u, f32(1) / (f32(2) - f32(2) * u)).astype(f32)
    t = (fu * f32(uniform_range) + f32(near)).astype(f32)
    return t[:-1].copy(), (t[1:] - t[:-1]).astype(f32)


def march_unbounded(rays_o, rays_d, n_samples, near=0.0, far=1e5, uniform_range=1.0):
    """reference core.py:47-59 (``far`` is unused there too)."""
    t, dl = unbounded_table(n_samples, near, uniform_range)
    R = np.asarray(rays_o).shape[0]
    return np.broadcast_to(t, (R, n_samples)), np.broadcast_to(dl, (R, n_samples))


# --------------------------------------------------------------------------- #
# a2 / a4: contractions
# --------------------------------------------------------------------------- #
def contract_aabb(coords, aabb):
    """reference core.py:26-31."""
    c = np.asarray(coords, f32)
    aabb = np.asarray(aabb, f32)
    mask = np.all((c >= aabb[0]) & (c <= aabb[1]), axis=-1)
    out = ((c - aabb[0]) / (aabb[1] - aabb[0]) * f32(2) - f32(1)).astype(f32)
    return out, mask


def contract_mip360(coords, order=float("inf")):
    """reference core.py:15-20."""
    c = np.asarray(coords, f32)
    if np.isinf(order):
        n = np.max(np.abs(c), axis=-1, keepdims=True)
    elif order == 2:
        sq = c * c
        n = np.sqrt((sq[..., 0:1] + sq[..., 1:2]) + sq[..., 2:3]).astype(f32)
    else:
        raise NotImplementedError(order)
    with np.errstate(divide="ignore", invalid="ignore"):
        far_branch = (f32(2) - f32(1) / n) * c / n
    out = (np.where(n <= f32(1), c, far_branch) / f32(2)).astype(f32)
    return out, None


# --------------------------------------------------------------------------- #
# a5 / a6: occupancy grid
# --------------------------------------------------------------------------- #
def trilinear_zeros_align(grid: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """ATen ``grid_sampler_3d`` (CPU), bilinear, zeros padding, align_corners=True, as
    called by reference core.py:151-155.  coords[...,0] indexes W (last grid dim),
    [...,1] H, [...,2] D.  Tap order tnw,tne,tsw,tse,bnw,bne,bsw,bse; each tap is
    one rounded multiply followed by one rounded add (verified bit-exact against
    torch 2.10 CPU in make_goldens.py)."""
    g = np.asarray(grid, f32)
    D, H, W = g.shape
    c = np.asarray(coords, f32).reshape(-1, 3)
    one, two = f32(1), f32(2)
    ix = ((c[:, 0] + one) / two) * f32(W - 1)
    iy = ((c[:, 1] + one) / two) * f32(H - 1)
    iz = ((c[:, 2] + one) / two) * f32(D - 1)
    x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
    x1, y1, z1 = x0 + one, y0 + one, z0 + one
    wx0, wx1 = x1 - ix, ix - x0
    wy0, wy1 = y1 - iy, iy - y0
    wz0, wz1 = z1 - iz, iz - z0
    taps = [
        (x0, y0, z0, wx0 * wy0 * wz0), (x1, y0, z0, wx1 * wy0 * wz0),
        (x0, y1, z0, wx0 * wy1 * wz0), (x1, y1, z0, wx1 * wy1 * wz0),
        (x0, y0, z1, wx0 * wy0 * wz1), (x1, y0, z1, wx1 * wy0 * wz1),
        (x0, y1, z1, wx0 * wy1 * wz1), (x1, y1, z1, wx1 * wy1 * wz1),
    ]
    out = np.zeros(c.shape[0], f32)
    for cx, cy, cz, wt in taps:
        ok = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H) & (cz >= 0) & (cz < D)
        xi = np.clip(np.nan_to_num(cx), 0, W - 1).astype(np.int64)
        yi = np.clip(np.nan_to_num(cy), 0, H - 1).astype(np.int64)
        zi = np.clip(np.nan_to_num(cz), 0, D - 1).astype(np.int64)
        out = (out + np.where(ok, g[zi, yi, xi] * wt.astype(f32), f32(0)).astype(f32)).astype(f32)
    return out.reshape(np.asarray(coords).shape[:-1])


def occupancy_threshold(base_threshold: float, mean: float) -> float:
    """reference core.py:125-127."""
    return min(base_threshold, mean)


def occupancy_query(grid, coords, threshold: float) -> np.ndarray:
    """reference core.py:147-156; the python-float threshold is compared in fp32."""
    return trilinear_zeros_align(grid, coords) > f32(threshold)


def occupancy_fraction(grid, threshold: float) -> float:
    """reference core.py:121-123."""
    g = np.asarray(grid, f32)
    return float((g > f32(threshold)).sum()) / g.size


def occupancy_voxel_coords(size: Sequence[int], i: int, jitter: np.ndarray) -> np.ndarray:
    """reference core.py:109-119,136: jittered centres of depth-slice ``i``.
    ``jitter`` is the U[0,1) tensor torch.rand_like(self.coords[i]) produced, shape
    [H,W,3]; coordinates come out ordered (x,y,z)=(w,h,d) because of the flip."""
    D, H, W = size
    hh, ww = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    ijk = np.stack([np.full_like(hh, f32(i)), hh, ww], -1)         # (d,h,w)
    flipped = ijk[..., ::-1]                                       # (w,h,d)
    sz = np.array(size, f32)                                       # NOT flipped (core.py:109,136)
    c = f32(-1) + f32(2) * (flipped + np.asarray(jitter, f32)) / sz
    return c.astype(f32).reshape(-1, 3)


def occupancy_update(grid, sigma_fn: Callable[[np.ndarray], np.ndarray], step_size, base_threshold,
                     decay, mean, jitters: Sequence[np.ndarray]):
    """reference core.py:133-145.  Returns (new_grid, new_mean)."""
    g = np.array(grid, f32, copy=True)
    size = g.shape
    thr = f32(occupancy_threshold(base_threshold, mean))     # constant during the sweep
    for i in range(size[0]):
        pts = occupancy_voxel_coords(size, i, jitters[i])
        sig = np.asarray(sigma_fn(pts), f32).reshape(size[1], size[2])
        alpha = (f32(1) - np.exp((-sig * f32(step_size)).astype(f32))).astype(f32)
        g[i] = np.where(alpha > thr, f32(1), (f32(decay) * g[i]).astype(f32))
    return g, float(torch_like_mean(g))


def torch_like_mean(g: np.ndarray) -> np.float32:
    """grid.mean() -- summation order is torch-internal; goldens compare with rtol."""
    return f32(np.mean(np.asarray(g, f32), dtype=np.float64))


# --------------------------------------------------------------------------- #
# a7 / a8: sample packing
# --------------------------------------------------------------------------- #
def ray_provider(rays_o, rays_d, *, marcher: str, contraction: str, grid, threshold: float,
                 n_samples: int, near: float = 0.0, far: float = 1e5, aabb=None,
                 uniform_range: float = 1.0, order=float("inf"), jitter: Optional[np.ndarray] = None):
    """reference core.py:165-188.  ``jitter`` (U[0,1) of shape [R,S]) stands in for
    torch.rand_like when training=True; None means training=False.
    Returns packed_samples [N,7] fp32, packing_info [R,2] int32."""
    o = np.asarray(rays_o, f32)
    d = np.asarray(rays_d, f32)
    if marcher == "aabb":
        t, dl = march_aabb(o, d, aabb, n_samples, near, far)
    else:
        t, dl = march_unbounded(o, d, n_samples, near, far, uniform_range)
    if jitter is not None:
        t = (t + np.asarray(jitter, f32) * dl).astype(f32)                       # core.py:173
    pts = (o[:, None, :] + (d[:, None, :] * t[..., None]).astype(f32)).astype(f32)  # core.py:174
    if contraction == "aabb":
        pts, mmask = contract_aabb(pts, aabb)
    else:
        pts, mmask = contract_mip360(pts, order)
    occ = occupancy_query(grid, pts, threshold)
    mask = occ if mmask is None else (mmask & occ)
    count = mask.sum(-1).astype(np.int32)                                        # core.py:179
    start = (np.cumsum(count, dtype=np.int32) - count).astype(np.int32)          # core.py:180
    info = np.stack([start, count], -1).astype(np.int32)
    packed_d = np.repeat(d, count, axis=0)
    packed_o = pts[mask]
    packed_dl = np.ascontiguousarray(dl)[mask]
    packed = np.concatenate([packed_o, packed_d, packed_dl[:, None]], -1).astype(f32)
    return packed, info


def uniform01(seed: int, ctr) -> np.ndarray:
    """The HIP sampler's counter-based U[0,1) (tinynerf_amd/csrc/tn_common.h ``tn::uniform01``: a splitmix64 finaliser over
    ``seed + golden * (ctr + 1)``, top 24 bits) restated with wrapping 64-bit integers.  It stands in for ``torch.rand_like``
    (reference core.py:172, :136), whose stream no other device reproduces either; restated here so that the PRODUCTION path of
    the sampler -- device RNG on, exit shortcut on -- can be held against ``ray_provider(..., jitter=...)`` bit for bit.
    Candidate (ray r, sample k) of an S-candidate pass draws counter r * S + k; voxel (slice, i) coordinate c of a refresh draws
    (slice * H * W + i) * 3 + c."""
    ctr = np.asarray(ctr, np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * (ctr + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.uint32).astype(f32) * f32(1.0 / 16777216.0)).astype(f32)


def sampler_jitter(seed: int, n_rays: int, n_samples: int) -> np.ndarray:
    """[R, S] jitter table of one sampler pass under device RNG ``seed`` (sampler.hip ``candidate``)"""
    ctr = np.arange(n_rays, dtype=np.uint64)[:, None] * np.uint64(n_samples) + np.arange(n_samples, dtype=np.uint64)[None, :]
    return uniform01(seed, ctr)


def dynamic_batch(loader_batches, provider: Callable, target_sample_size: int):
    """reference run.py:215-244 -- accumulate loader batches until the projection rule
    trips.  ``loader_batches`` yields (rays_o, rays_d, rgbs); ``provider`` maps
    (o,d)->(packed,info).  Returns packed, info, rgbs, k."""
    cur, proj, k = 0, 0, 0
    accs, acci, accr = [], [], []
    it = iter(loader_batches)
    while proj < target_sample_size:
        o, d, c = next(it)
        p, info = provider(o, d)
        info = info.copy()
        info[:, 0] += np.int32(cur)
        accs.append(p); acci.append(info); accr.append(c)
        cur += p.shape[0]
        k += 1
        proj = int(cur * (1 + 1 / k))
    return np.concatenate(accs, 0), np.concatenate(acci, 0), np.concatenate(accr, 0), k


# --------------------------------------------------------------------------- #
# a9 - a14: encodings and MLP heads
# --------------------------------------------------------------------------- #
def posenc_freqs(n_freqs: int) -> np.ndarray:
    """reference models.py:34 -- 2**j * pi as fp32."""
    return (np.float64(2.0) ** np.arange(n_freqs) * np.pi).astype(f32)


def posenc(x, n_freqs: int) -> np.ndarray:
    """reference models.py:36-39: per coordinate [sin(x f_0..), cos(x f_0..)]."""
    x = np.asarray(x, f32)
    ang = (x[..., None] * posenc_freqs(n_freqs)).astype(f32)
    enc = np.concatenate([np.sin(ang), np.cos(ang)], -1).astype(f32)
    return enc.reshape(*x.shape[:-1], x.shape[-1] * 2 * n_freqs)


def mlp_layers(sd: Dict[str, np.ndarray], prefix: str) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Collect (weight, bias) pairs of a reference ``MLP`` (models.py:18-26) from a
    state_dict: keys ``net.0``, ``net.{2..L+1}.0``, ``net.{L+2}``."""
    keys = sorted({k[len(prefix):].rsplit(".", 1)[0] for k in sd
                   if k.startswith(prefix) and k.endswith(".weight")},
                  key=lambda s: int(s.split(".")[0]))
    return [(np.asarray(sd[prefix + k + ".weight"], f32), np.asarray(sd[prefix + k + ".bias"], f32))
            for k in keys]


def mlp_forward(x, layers: Sequence[Tuple[np.ndarray, np.ndarray]], return_hidden=False):
    """reference models.py:7-28: ReLU after every layer but the last."""
    h = np.asarray(x, f32)
    hidden = [h]
    for li, (w, b) in enumerate(layers):
        h = (h @ w.T + b).astype(f32)
        if li + 1 < len(layers):
            h = np.maximum(h, f32(0))
        hidden.append(h)
    return (h, hidden) if return_hidden else h


def sigma_decoder(feat, layers):
    """reference models.py:70-77: exp(mlp(feat) - 1)."""
    y = mlp_forward(feat, layers)
    return np.exp((y - f32(1)).astype(f32)).astype(f32)


def color_decoder(feat, dirs, layers, n_freqs: int):
    """reference models.py:79-89: sigmoid(mlp(cat[PE(d), d, feat]))."""
    x = np.concatenate([posenc(dirs, n_freqs), np.asarray(dirs, f32), np.asarray(feat, f32)], -1)
    y = mlp_forward(x, layers)
    return (f32(1) / (f32(1) + np.exp(-y))).astype(f32)


def vanilla_features(x, layers, n_freqs: int):
    """reference models.py:59-68."""
    return mlp_forward(posenc(x, n_freqs), layers)


# --------------------------------------------------------------------------- #
# a15 / a16 / a19: K-Planes
# --------------------------------------------------------------------------- #
def bilinear_zeros_align(plane: np.ndarray, xy: np.ndarray) -> np.ndarray:
    """ATen ``grid_sampler_2d`` bilinear, zeros padding, align_corners=True as used by
    reference models.py:108-112.  plane [C,H,W]; xy[...,0] indexes W, [...,1] H.
    Returns [N,C]."""
    p = np.asarray(plane, f32)
    C, H, W = p.shape
    c = np.asarray(xy, f32).reshape(-1, 2)
    one, two = f32(1), f32(2)
    ix = ((c[:, 0] + one) / two) * f32(W - 1)
    iy = ((c[:, 1] + one) / two) * f32(H - 1)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + one, y0 + one
    taps = [(x0, y0, (x1 - ix) * (y1 - iy)), (x1, y0, (ix - x0) * (y1 - iy)),
            (x0, y1, (x1 - ix) * (iy - y0)), (x1, y1, (ix - x0) * (iy - y0))]
    out = np.zeros((c.shape[0], C), f32)
    for cx, cy, wt in taps:
        ok = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
        xi = np.clip(np.nan_to_num(cx), 0, W - 1).astype(np.int64)
        yi = np.clip(np.nan_to_num(cy), 0, H - 1).astype(np.int64)
        v = p[:, yi, xi].T                                           # [N,C]
        out = out + np.where(ok[:, None], v * wt[:, None].astype(f32), f32(0)).astype(f32)
    return out


KPLANES_PAIRS = list(itertools.combinations(range(3), 2))    # models.py:146 -> (0,1),(0,2),(1,2)


def kplanes_features(x, planes: Sequence[Sequence[np.ndarray]]):
    """reference models.py:153-163.  planes[scale][pair] is [1,C,H,W] or [C,H,W]."""
    x = np.asarray(x, f32)
    feats = []
    for scale in planes:
        prod = None
        for (i, j), pl in zip(KPLANES_PAIRS, scale):
            pl = np.asarray(pl, f32)
            pl = pl[0] if pl.ndim == 4 else pl
            v = bilinear_zeros_align(pl, x[..., (i, j)])
            prod = v if prod is None else (prod * v).astype(f32)
        feats.append(prod)
    return np.concatenate(feats, -1).astype(f32)


def plane_loss_tv(plane) -> np.float64:
    """reference models.py:115-118 (mse over both spatial differences)."""
    p = np.asarray(plane, np.float64)
    return np.mean((p[:, :, 1:, :] - p[:, :, :-1, :]) ** 2) + np.mean((p[:, :, :, 1:] - p[:, :, :, :-1]) ** 2)


def plane_loss_l1(plane) -> np.float64:
    """reference models.py:120-121."""
    return np.mean(np.abs(np.asarray(plane, np.float64)))


def kplanes_loss_tv(planes) -> float:
    """reference models.py:165-172."""
    v = [plane_loss_tv(p) for s in planes for p in s]
    return float(sum(v) / len(v))


def kplanes_loss_l1(planes) -> float:
    """reference models.py:174-181."""
    v = [plane_loss_l1(p) for s in planes for p in s]
    return float(sum(v) / len(v))


# --------------------------------------------------------------------------- #
# a20: Cobafa
# --------------------------------------------------------------------------- #
def trilinear_channels(grid: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """grid [C,D,H,W] sampled like reference models.py:232-236 -> [N,C]."""
    g = np.asarray(grid, f32)
    return np.stack([trilinear_zeros_align(g[c], coords) for c in range(g.shape[0])], -1)


def cobafa_gather(x, basis_grids, coef_grid, freqs):
    """reference models.py:259-265: the concatenated basis x coefficient features, before dropout + MLP."""
    x = np.asarray(x, f32)
    coefs = trilinear_channels(np.asarray(coef_grid, f32)[0], x)
    feats = []
    for i, (fr, bg) in enumerate(zip(freqs, basis_grids)):
        saw = (f32(2) * np.mod((f32(fr) * x).astype(f32), f32(1)) - f32(1)).astype(f32)   # models.py:213
        feats.append(trilinear_channels(np.asarray(bg, f32)[0], saw) * coefs[:, [i]])
    return np.concatenate(feats, -1).astype(f32)


def cobafa_features(x, basis_grids, coef_grid, freqs, mlp):
    """reference models.py:257-266 in eval mode (Dropout(0.01) inactive)."""
    x = np.asarray(x, f32)
    coefs = trilinear_channels(np.asarray(coef_grid, f32)[0], x)
    feats = []
    for i, (fr, bg) in enumerate(zip(freqs, basis_grids)):
        saw = (f32(2) * np.mod((f32(fr) * x).astype(f32), f32(1)) - f32(1)).astype(f32)   # models.py:213
        feats.append(trilinear_channels(np.asarray(bg, f32)[0], saw) * coefs[:, [i]])
    return mlp_forward(np.concatenate(feats, -1).astype(f32), mlp)


# --------------------------------------------------------------------------- #
# a17: NeRF weights (restatement of the native kernels)
# --------------------------------------------------------------------------- #
def weights_fwd_py(sigmas, steps, info, threshold: float) -> np.ndarray:
    """reference cuda.cu:14-28, python loops (small inputs only).  T is fp32, the
    product T*(1-alpha) is evaluated in fp64 and rounded to fp32 (cuda.cu:25)."""
    s = np.asarray(sigmas, f32); dl = np.asarray(steps, f32); info = np.asarray(info, np.int32)
    w = np.zeros_like(s)
    thr = f32(threshold)
    for start, cnt in info:
        T = f32(1)
        k, end = int(start), int(start) + int(cnt)
        while T > thr and k < end:
            a = f32(np.exp(f32(-s[k] * dl[k])))
            w[k] = f32(np.float64(T) * (1.0 - np.float64(a)))
            T = f32(T * a)
            k += 1
    return w


def weights_bwd_py(sigmas, steps, info, weights, grad_w) -> np.ndarray:
    """reference cuda.cu:49-56, python loops.  No early termination in backward."""
    s = np.asarray(sigmas, f32); dl = np.asarray(steps, f32); info = np.asarray(info, np.int32)
    w = np.asarray(weights, f32); g = np.asarray(grad_w, f32)
    out = np.zeros_like(s)
    for start, cnt in info:
        a0, a1 = int(start), int(start) + int(cnt)
        acc = f32(0)
        for k in range(a0, a1):
            acc = f32(acc - f32(w[k] * g[k]))
        T = f32(1)
        for k in range(a0, a1):
            acc = f32(acc + f32(w[k] * g[k]))
            T = f32(T * f32(np.exp(f32(-s[k] * dl[k]))))
            out[k] = f32(dl[k] * f32(acc + f32(T * g[k])))
    return out


def weights_bwd_fp64(sigmas, steps, info, weights, grad_w) -> np.ndarray:
    """The formula of cuda.cu:49-56 evaluated exactly (fp64, suffix sums formed directly instead of `-total + prefix`).
    The reference's fp32 version cancels catastrophically behind a terminated ray (acc = -sum + prefix is rounding noise
    of size eps * |sum| where the true suffix sum is ~0) and multiplies that noise by the step size, which is up to
    ~10 in unbounded scenes: its result can be percent-level away from this one.  Tests use the difference between the
    two as the conditioning of a fixture, i.e. as the tolerance any other fp32 evaluation order is entitled to."""
    s = np.asarray(sigmas, np.float64); dl = np.asarray(steps, np.float64)
    w = np.asarray(weights, np.float64); g = np.asarray(grad_w, np.float64)
    out = np.zeros_like(s)
    for start, cnt in np.asarray(info):
        sl = slice(int(start), int(start) + int(cnt))
        wg = w[sl] * g[sl]
        suffix = np.concatenate([np.cumsum(wg[::-1])[::-1][1:], [0.0]])
        T = np.cumprod(np.exp(-s[sl] * dl[sl]))
        out[sl] = dl[sl] * (T * g[sl] - suffix)
    return out.astype(f32)


def weights_fwd_vectorised(sigmas, steps, info, threshold: float) -> np.ndarray:
    """Independent formulation used to cross-check the loop restatement:
    alpha=1-exp(-sigma*delta); T=exclusive cumprod per ray; w=T*alpha where T>thr."""
    s = np.asarray(sigmas, np.float64); dl = np.asarray(steps, np.float64)
    w = np.zeros_like(s)
    for start, cnt in np.asarray(info):
        sl = slice(int(start), int(start) + int(cnt))
        a = np.exp(-s[sl] * dl[sl])
        T = np.concatenate([[1.0], np.cumprod(a)[:-1]])
        w[sl] = np.where(T > threshold, T * (1 - a), 0.0)
    return w


_WEIGHTS_LIB = None


def _weights_lib():
    """Load (building if needed) the C twin oracle/weights_ref.c."""
    global _WEIGHTS_LIB
    if _WEIGHTS_LIB is None:
        so = os.path.join(_HERE, "_build", "libweights_ref.so")
        if not os.path.exists(so):
            import subprocess
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                                   os.path.join(_HERE, "weights_ref.c"), "-o", so, "-lm"])
        lib = ctypes.CDLL(so)
        fp = ctypes.POINTER(ctypes.c_float); ip = ctypes.POINTER(ctypes.c_int32)
        lib.oracle_weights_fwd.argtypes = [fp, fp, ip, ctypes.c_float, fp, ctypes.c_int64]
        lib.oracle_weights_bwd.argtypes = [fp, fp, ip, fp, fp, fp, ctypes.c_int64]
        lib.oracle_composite.argtypes = [fp, fp, ip, fp, fp, fp, ctypes.c_int64, ctypes.c_int]
        _WEIGHTS_LIB = lib
    return _WEIGHTS_LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def weights_fwd(sigmas, steps, info, threshold: float) -> np.ndarray:
    """C restatement of cuda.cu:14-28 (oracle/weights_ref.c)."""
    s = np.ascontiguousarray(sigmas, f32); dl = np.ascontiguousarray(steps, f32)
    info = np.ascontiguousarray(info, np.int32)
    w = np.zeros_like(s)
    _weights_lib().oracle_weights_fwd(_fp(s), _fp(dl), _ip(info), ctypes.c_float(threshold), _fp(w), info.shape[0])
    return w


def weights_bwd(sigmas, steps, info, weights, grad_w) -> np.ndarray:
    """C restatement of cuda.cu:49-56 (oracle/weights_ref.c)."""
    s = np.ascontiguousarray(sigmas, f32); dl = np.ascontiguousarray(steps, f32)
    info = np.ascontiguousarray(info, np.int32)
    w = np.ascontiguousarray(weights, f32); g = np.ascontiguousarray(grad_w, f32)
    out = np.zeros_like(s)
    _weights_lib().oracle_weights_bwd(_fp(s), _fp(dl), _ip(info), _fp(w), _fp(g), _fp(out), info.shape[0])
    return out


# --------------------------------------------------------------------------- #
# a18: renderer
# --------------------------------------------------------------------------- #
def composite(rgbs, weights, info, bg_color=None) -> np.ndarray:
    """reference core.py:256-265: per-ray sum of w*rgb (+ bg*(1-sum w)).  Summation is
    sequential in sample order, like index_add_ on CPU."""
    rgbs = np.asarray(rgbs, f32); w = np.asarray(weights, f32); info = np.asarray(info, np.int32)
    R = info.shape[0]
    out = np.zeros((R, 3), f32)
    opac = np.zeros(R, f32)
    for r, (start, cnt) in enumerate(info):
        for k in range(int(start), int(start) + int(cnt)):
            out[r] = out[r] + (rgbs[k] * w[k]).astype(f32)
            opac[r] = f32(opac[r] + w[k])
    if bg_color is not None:
        out = (out + np.asarray(bg_color, f32)[None] * (f32(1) - opac[:, None])).astype(f32)
    return out


def render(packed, info, feature_fn, sigma_fn, rgb_fn, bg_color=None, threshold: float = 1e-4):
    """reference core.py:225-267 including the empty-iteration branch."""
    packed = np.asarray(packed, f32); info = np.asarray(info, np.int32)
    n = packed.shape[0]
    rgbs = np.zeros((n, 3), f32)
    w = np.zeros(n, f32)
    if n > 0:
        feat = feature_fn(packed[:, :3])
        sig = np.asarray(sigma_fn(feat), f32).ravel()
        w = weights_fwd(sig, packed[:, 6], info, threshold)
        m = w > 0
        if m.any():
            rgbs[m] = rgb_fn(feat[m], packed[:, 3:6][m])
        else:
            w = np.zeros(n, f32)
    return composite(rgbs, w, info, bg_color)


def psnr(x, y) -> float:
    """reference run.py:53-54."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    return float(-10.0 * np.log10(np.mean((x - y) ** 2)))


# --------------------------------------------------------------------------- #
# ray generation (next row f-2, used for synthetic bench inputs)
# --------------------------------------------------------------------------- #
def generate_rays(camera: np.ndarray, fx, fy, cx, cy, w: int, h: int):
    """reference data.py:48-73 for one camera [4,4] -> rays_o, rays_d of shape [h,w,3]."""
    cam = np.asarray(camera, f32)
    xs, ys = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32), indexing="xy")
    g = np.stack([xs, ys], -1)
    g = ((g - np.array([cx, cy], f32) + f32(0.5)) / np.array([fx, -fy], f32)).astype(f32)
    g = np.concatenate([g, np.full(g.shape[:-1] + (1,), f32(-1))], -1)
    d = (g @ cam[:3, :3].T).astype(f32)
    d = d / np.sqrt((d * d).sum(-1, keepdims=True)).astype(f32)
    o = np.broadcast_to(cam[:3, 3], d.shape)
    return o.astype(f32), d.astype(f32)
