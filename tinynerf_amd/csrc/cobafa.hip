// Cobafa factorised feature field (reference src/models.py:209-266): per sample one trilinear lookup into
// the coefficient grid (L channels) and, per level, one trilinear lookup of the sawtooth-warped point into a
// basis grid; features = basis * coefficient, levels concatenated.  Replaces 7 x grid_sampler_3d + transposes
// + muls + cat of the reference with one launch each way.  Channel-last grids: a voxel's channels are one
// 16-32 B run.  Thread = sample (36 outputs); the backward scatters with fp32 atomics.
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

struct Taps3 {
    int off[8];      // voxel index (z*H + y)*W + x, or -1 when out of bounds
    float w[8];
};

// ATen grid_sampler_3d conventions (align_corners=True, zeros padding); p = (x -> W, y -> H, z -> D)
__device__ __forceinline__ Taps3 taps3(const float p[3], int D, int H, int W) {
    Taps3 t;
    const float ix = ((p[0] + 1.0f) * 0.5f) * (float)(W - 1);
    const float iy = ((p[1] + 1.0f) * 0.5f) * (float)(H - 1);
    const float iz = ((p[2] + 1.0f) * 0.5f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
    const float gx = (x0 + 1.0f) - ix, gy = (y0 + 1.0f) - iy, gz = (z0 + 1.0f) - iz;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float cx = x0 + (float)(k & 1), cy = y0 + (float)((k >> 1) & 1), cz = z0 + (float)(k >> 2);
        const bool ok = cx >= 0.0f && cx < (float)W && cy >= 0.0f && cy < (float)H && cz >= 0.0f && cz < (float)D;
        t.off[k] = ok ? ((int)cz * H + (int)cy) * W + (int)cx : -1;
        t.w[k] = ((k & 1) ? fx : gx) * (((k >> 1) & 1) ? fy : gy) * ((k >> 2) ? fz : gz);
    }
    return t;
}

__device__ __forceinline__ void sawtooth(const float x[3], float f, float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = f * x[c];
        const float s = 2.0f * (v - floorf(v)) - 1.0f;  // torch: (f*x) % 1 in [0,1)  (models.py:213)
        y[c] = f > 0.0f ? s : x[c];                     // f <= 0: plain grid lookup (CobafaGrid on its own)
    }
}

__global__ __launch_bounds__(256) void cobafa_fwd_kernel(CbArgs a, const float *__restrict__ x, int64_t n, float *__restrict__ feat)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    float coef[TN_COBAFA_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) coef[l] = 0.0f;
    {
        const Taps3 t = taps3(p, a.cres[0], a.cres[1], a.cres[2]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t.off[k] < 0) continue;
            const float *v = a.coef + (int64_t)t.off[k] * a.n_levels;
#pragma unroll
            for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l)
                if (l < a.n_levels) coef[l] += v[l] * t.w[k];
        }
    }
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
        if (l >= a.n_levels) break;
        float y[3];
        sawtooth(p, a.freq[l], y);
        const Taps3 t = taps3(y, a.res[l][0], a.res[l][1], a.res[l][2]);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int C = a.ch[l];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t.off[k] < 0) continue;
            const float *v = a.basis[l] + (int64_t)t.off[k] * C;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) acc[c] += v[c] * t.w[k];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) feat[i * a.feat_dim + a.off[l] + c] = acc[c] * coef[l];
    }
}

__global__ __launch_bounds__(256) void cobafa_bwd_kernel(CbArgs a, const float *__restrict__ x, int64_t n,
                                                         const float *__restrict__ gfeat)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    const Taps3 tc = taps3(p, a.cres[0], a.cres[1], a.cres[2]);
    float coef[TN_COBAFA_MAX_LEVELS], gcoef[TN_COBAFA_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) { coef[l] = 0.0f; gcoef[l] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (tc.off[k] < 0) continue;
        const float *v = a.coef + (int64_t)tc.off[k] * a.n_levels;
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l)
            if (l < a.n_levels) coef[l] += v[l] * tc.w[k];
    }
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
        if (l >= a.n_levels) break;
        float y[3];
        sawtooth(p, a.freq[l], y);
        const Taps3 t = taps3(y, a.res[l][0], a.res[l][1], a.res[l][2]);
        const int C = a.ch[l];
        float g[8], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = c < C ? gfeat[i * a.feat_dim + a.off[l] + c] : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t.off[k] < 0) continue;
            const float *v = a.basis[l] + (int64_t)t.off[k] * C;
            float *gv = a.gbasis[l] + (int64_t)t.off[k] * C;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) {
                    acc[c] += v[c] * t.w[k];
                    atomicAdd(gv + c, g[c] * coef[l] * t.w[k]);      // d feat / d basis = coef * w
                }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) gcoef[l] += g[c] * acc[c];        // d feat / d coef = basis value
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (tc.off[k] < 0) continue;
        float *gv = a.gcoef + (int64_t)tc.off[k] * a.n_levels;
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l)
            if (l < a.n_levels) atomicAdd(gv + l, gcoef[l] * tc.w[k]);
    }
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
    cobafa_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a, x, n, grad_feat);
    return tn::check_launch("cobafa_bwd_kernel");
}
