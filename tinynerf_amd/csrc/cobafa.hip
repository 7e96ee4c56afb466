// Cobafa factorised feature field (reference src/models.py:209-266): per sample one trilinear lookup into
// the coefficient grid (L channels) and, per level, one trilinear lookup of the sawtooth-warped point into a
// basis grid; features = basis * coefficient, levels concatenated.  Replaces 7 x grid_sampler_3d + transposes
// + muls + cat of the reference with one launch each way.  Channel-last grids: a voxel's channels are one
// 16-32 B run.  Forward: thread = sample (36 outputs); backward: transposed scatter (see below).
#include "tn_common.h"
#include <algorithm>

namespace {

struct CbArgs {
    int n_levels, feat_dim;
    int cres[3];
    int res[TN_COBAFA_MAX_LEVELS][3];
    int ch[TN_COBAFA_MAX_LEVELS], off[TN_COBAFA_MAX_LEVELS];
    float freq[TN_COBAFA_MAX_LEVELS];
    const float *coef;
    const float *basis[TN_COBAFA_MAX_LEVELS];
    float *gcoef;
    float *gbasis[TN_COBAFA_MAX_LEVELS];
};

__device__ __forceinline__ void sawtooth(const float x[3], float f, float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = f * x[c];
        const float s = 2.0f * (v - floorf(v)) - 1.0f;  // torch: (f*x) % 1 in [0,1)  (models.py:213)
        y[c] = f > 0.0f ? s : x[c];                     // f <= 0: plain grid lookup (CobafaGrid on its own)
    }
}

// ATen grid_sampler_3d conventions (align_corners=True, zeros padding); p = (x -> W, y -> H, z -> D)
struct Cell3 {
    int base, mask;                    // voxel index of the (x0,y0,z0) corner; bit k: tap k is inside the grid
    float fx, gx, fy, gy, fz, gz;
};

__device__ __forceinline__ Cell3 cell3(const float p[3], int D, int H, int W) {
    Cell3 t;
    const float ix = ((p[0] + 1.0f) * 0.5f) * (float)(W - 1);
    const float iy = ((p[1] + 1.0f) * 0.5f) * (float)(H - 1);
    const float iz = ((p[2] + 1.0f) * 0.5f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    t.fx = ix - x0; t.fy = iy - y0; t.fz = iz - z0;
    t.gx = (x0 + 1.0f) - ix; t.gy = (y0 + 1.0f) - iy; t.gz = (z0 + 1.0f) - iz;
    int m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float cx = x0 + (float)(k & 1), cy = y0 + (float)((k >> 1) & 1), cz = z0 + (float)(k >> 2);
        const bool ok = cx >= 0.0f && cx < (float)W && cy >= 0.0f && cy < (float)H && cz >= 0.0f && cz < (float)D;
        m |= ok ? (1 << k) : 0;
    }
    t.mask = m;
    t.base = m ? ((int)z0 * H + (int)y0) * W + (int)x0 : 0;
    return t;
}
__device__ __forceinline__ float tap_weight(const Cell3 &t, int k) {
    return ((k & 1) ? t.fx : t.gx) * (((k >> 1) & 1) ? t.fy : t.gy) * ((k >> 2) ? t.fz : t.gz);
}
__device__ __forceinline__ int tap_delta(int k, int H, int W) { return (k & 1) + ((k >> 1) & 1) * W + (k >> 2) * W * H; }

// lane = sample: sum_k w_k * grid[tap k][c]; taps outside the grid are read from voxel 0 with weight 0
__device__ __forceinline__ void gather8(const Cell3 &t, const float *__restrict__ grid, int C, int H, int W, float acc[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool ok = (t.mask >> k) & 1;
        const float *v = grid + (int64_t)(ok ? t.base + tap_delta(k, H, W) : 0) * C;
        const float w = ok ? tap_weight(t, k) : 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) acc[c] += v[c] * w;
    }
}

__global__ __launch_bounds__(256) void cobafa_fwd_kernel(CbArgs a, const float *__restrict__ x, int64_t n, float *__restrict__ feat)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    float coef[8];
    gather8(cell3(p, a.cres[0], a.cres[1], a.cres[2]), a.coef, a.n_levels, a.cres[1], a.cres[2], coef);
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
        if (l >= a.n_levels) break;
        float y[3], acc[8];
        sawtooth(p, a.freq[l], y);
        const int C = a.ch[l];
        gather8(cell3(y, a.res[l][0], a.res[l][1], a.res[l][2]), a.basis[l], C, a.res[l][1], a.res[l][2], acc);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) feat[i * a.feat_dim + a.off[l] + c] = acc[c] * coef[l];
    }
}

// ------------------------------------------------------------------------------------------------
// backward: transposed scatter.  A lane-per-sample scatter issues wave atomics whose 64 lanes hit 64 different lines
// (~20 G lane-atomics/s on the chip: 17.7 ms for 2^20 samples x 8 taps x 42 channels).  Here a wave does the per-sample
// arithmetic with lane = sample (phase A: cell, fractions, the forward values the coefficient gradient needs), publishes
// g * coef through a wave-private LDS tile, and then walks its samples with lane = (tap, channel) (phase B): the per-sample
// cell / fractions travel as scalars (v_readlane), the two x-adjacent taps' channels are one contiguous 32-64 B run, so an
// atomic instruction is 4-8 memory requests instead of 64.  Levels with <= 4 channels take two samples per instruction.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rl(int v, int s) { return __builtin_amdgcn_readlane(v, s); }
__device__ __forceinline__ float rl(float v, int s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s)); }

// phase B of one grid: tile[s * 8 + c] holds the per-sample, per-channel factor; every tap receives factor * w_tap
__device__ __forceinline__ void scatter_grid(const float *tile, const Cell3 &t, int C, int H, int W, float *__restrict__ grad, int cnt, int lane)
{
    if (C > 4) {
        const int k = lane >> 3, c = lane & 7;
        const int dk = tap_delta(k, H, W);
#pragma unroll 4
        for (int s = 0; s < cnt; ++s) {
            Cell3 u;
            u.base = rl(t.base, s); u.mask = rl(t.mask, s);
            u.fx = rl(t.fx, s); u.gx = rl(t.gx, s); u.fy = rl(t.fy, s); u.gy = rl(t.gy, s); u.fz = rl(t.fz, s); u.gz = rl(t.gz, s);
            const float v = tile[s * 8 + c] * tap_weight(u, k);
            if (((u.mask >> k) & 1) && c < C) atomicAdd(grad + (int64_t)(u.base + dk) * C + c, v);
        }
    } else {
        const int half = lane >> 5, k = (lane >> 2) & 7, c = lane & 3;
        const int dk = tap_delta(k, H, W);
#pragma unroll 2
        for (int s = 0; s < cnt; s += 2) {
            const int s1 = s + 1 < 64 ? s + 1 : 63;
            Cell3 u;
            u.base = half ? rl(t.base, s1) : rl(t.base, s); u.mask = half ? rl(t.mask, s1) : rl(t.mask, s);
            u.fx = half ? rl(t.fx, s1) : rl(t.fx, s); u.gx = half ? rl(t.gx, s1) : rl(t.gx, s);
            u.fy = half ? rl(t.fy, s1) : rl(t.fy, s); u.gy = half ? rl(t.gy, s1) : rl(t.gy, s);
            u.fz = half ? rl(t.fz, s1) : rl(t.fz, s); u.gz = half ? rl(t.gz, s1) : rl(t.gz, s);
            const int ss = s + half;
            const float v = tile[(ss < 64 ? ss : 63) * 8 + c] * tap_weight(u, k);
            if (ss < cnt && ((u.mask >> k) & 1) && c < C) atomicAdd(grad + (int64_t)(u.base + dk) * C + c, v);
        }
    }
}

constexpr int CB_WAVES = 4;

__global__ __launch_bounds__(CB_WAVES * 64) void cobafa_bwd_kernel(CbArgs a, const float *__restrict__ x, int64_t n,
                                                                   const float *__restrict__ gfeat)
{
    __shared__ __attribute__((aligned(16))) float lds[CB_WAVES][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = lds[wave];
    const int64_t first = ((int64_t)blockIdx.x * CB_WAVES + wave) * 64;
    if (first >= n) return;                                     // wave-uniform
    const int cnt = (int)(n - first < 64 ? n - first : 64);
    const bool valid = lane < cnt;
    const int64_t i = valid ? first + lane : n - 1;
    const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    Cell3 tc = cell3(p, a.cres[0], a.cres[1], a.cres[2]);
    if (!valid) tc.mask = 0;
    float coef[8], gcoef[TN_COBAFA_MAX_LEVELS];
    gather8(tc, a.coef, a.n_levels, a.cres[1], a.cres[2], coef);
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) gcoef[l] = 0.0f;
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
        if (l >= a.n_levels) break;
        float y[3];
        sawtooth(p, a.freq[l], y);
        Cell3 t = cell3(y, a.res[l][0], a.res[l][1], a.res[l][2]);
        if (!valid) t.mask = 0;
        const int C = a.ch[l], H = a.res[l][1], W = a.res[l][2];
        float g[8], acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = gfeat[i * a.feat_dim + a.off[l] + (c < C ? c : 0)];
        gather8(t, a.basis[l], C, H, W, acc);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= C) g[c] = 0.0f;
            gcoef[l] += g[c] * acc[c];                          // d feat / d coef = basis value
            tile[lane * 8 + c] = g[c] * coef[l];                // d feat / d basis = coef * w
        }
        scatter_grid(tile, t, C, H, W, a.gbasis[l], cnt, lane);
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) tile[lane * 8 + l] = l < TN_COBAFA_MAX_LEVELS ? gcoef[l] : 0.0f;
    scatter_grid(tile, tc, a.n_levels, a.cres[1], a.cres[2], a.gcoef, cnt, lane);
}

int make_args(const tn_cobafa_desc *d, CbArgs &a)
{
    TN_REQUIRE(d, TN_E_NULL, "cobafa: null descriptor");
    TN_REQUIRE(d->n_levels >= 1 && d->n_levels <= TN_COBAFA_MAX_LEVELS, TN_E_CONFIG, "cobafa: n_levels out of range");
    TN_REQUIRE(d->coef, TN_E_NULL, "cobafa: null coefficient grid");
    a.n_levels = d->n_levels; a.coef = d->coef; a.gcoef = nullptr;
    int off = 0;
    for (int c = 0; c < 3; ++c) { TN_REQUIRE(d->coef_res[c] > 0, TN_E_SIZE, "cobafa: bad coef resolution"); a.cres[c] = d->coef_res[c]; }
    for (int l = 0; l < d->n_levels; ++l) {
        TN_REQUIRE(d->basis[l], TN_E_NULL, "cobafa: null basis grid");
        TN_REQUIRE(d->channels[l] >= 1 && d->channels[l] <= 8, TN_E_CONFIG, "cobafa: channels must be in [1, 8]");
        for (int c = 0; c < 3; ++c) { TN_REQUIRE(d->res[l][c] > 0, TN_E_SIZE, "cobafa: bad basis resolution"); a.res[l][c] = d->res[l][c]; }
        TN_REQUIRE((int64_t)d->res[l][0] * d->res[l][1] * d->res[l][2] < (1ll << 31), TN_E_SIZE, "cobafa: grid too large");
        a.ch[l] = d->channels[l]; a.off[l] = off; off += d->channels[l];
        a.freq[l] = d->freqs[l]; a.basis[l] = d->basis[l]; a.gbasis[l] = nullptr;
    }
    a.feat_dim = off;
    return TN_OK;
}

}  // namespace

extern "C" int tn_cobafa_fwd(const tn_cobafa_desc *desc, const float *x, int64_t n, float *feat, void *stream)
{
    CbArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_cobafa_fwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && feat, TN_E_NULL, "tn_cobafa_fwd: null pointer");
    cobafa_fwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a, x, n, feat);
    return tn::check_launch("cobafa_fwd_kernel");
}

extern "C" int tn_cobafa_bwd(const tn_cobafa_desc *desc, const float *x, int64_t n, const float *grad_feat, float *grad_coef,
                             float *const *grad_basis, void *stream)
{
    CbArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_cobafa_bwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && grad_feat && grad_coef && grad_basis, TN_E_NULL, "tn_cobafa_bwd: null pointer");
    a.gcoef = grad_coef;
    for (int l = 0; l < a.n_levels; ++l) {
        TN_REQUIRE(grad_basis[l], TN_E_NULL, "tn_cobafa_bwd: null basis gradient");
        a.gbasis[l] = grad_basis[l];
    }
    cobafa_bwd_kernel<<<dim3((unsigned)((n + 64 * CB_WAVES - 1) / (64 * CB_WAVES))), dim3(64 * CB_WAVES), 0, (hipStream_t)stream>>>(a, x, n, grad_feat);
    return tn::check_launch("cobafa_bwd_kernel");
}
