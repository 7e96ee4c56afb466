// K-Planes feature field (reference src/models.py:93-163): 3 scales x 3 planes of bilinear
// lookups, Hadamard product over the planes of a scale, scales concatenated.
// Replaces 9 x (grid_sampler_2d + transpose + contiguous) + 6 muls + cat of the reference
// (27+ launches) with one launch; backward scatters with hardware fp32 atomics.
#include "kplanes_scatter.h"
#include <algorithm>

namespace {

using tn::f32x4k;

struct KpArgs {
    int n_scales, C;
    int H[TN_KPLANES_MAX_SCALES], W[TN_KPLANES_MAX_SCALES];
    const float *planes[TN_KPLANES_MAX_SCALES][3];
    float *grads[TN_KPLANES_MAX_SCALES][3];
};

// NV = C/8 float4 groups per lane (C = 32 -> 4)
template <int NV>
__global__ __launch_bounds__(256) void kplanes_fwd_kernel(KpArgs a, const float *__restrict__ x, int64_t x_stride,
                                                          int64_t n, float *__restrict__ feat)
{
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 31) >> 5;
    const int C = a.C, FD = a.n_scales * C;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + j;
        if (row >= n) continue;
        const float xs[3] = {x[row * x_stride], x[row * x_stride + 1], x[row * x_stride + 2]};
        for (int s = 0; s < a.n_scales; ++s) {
            f32x4k prod[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) prod[q] = f32x4k{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (a.planes[s][p] == nullptr) continue;        // absent plane == factor 1 (single-plane lookups)
                float u, v;
                tn::pair_uv(xs, p, u, v);
                const tn::PlaneTaps t = tn::plane_taps(u, v, a.H[s], a.W[s], C);
                f32x4k val[NV];
                tn::plane_gather<NV>(a.planes[s][p], t, h * (C / 2), val);
#pragma unroll
                for (int q = 0; q < NV; ++q) prod[q] = prod[q] * val[q];      // (1*p0)*p1*p2, models.py:157-160
            }
            f32x4k *o = reinterpret_cast<f32x4k *>(feat + row * FD + s * C + h * (C / 2));
#pragma unroll
            for (int q = 0; q < NV; ++q) o[q] = prod[q];
        }
    }
}

// ---- backward: transposed, run-merged scatter (kplanes_scatter.h), one scale after the other ----
template <int NV>
__global__ __launch_bounds__(256, 3) void kplanes_bwd_kernel(KpArgs a, const float *__restrict__ x, int64_t x_stride,
                                                          int64_t n, const float *__restrict__ grad_feat)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * tn::KP_WAVE_LDS];
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    float *wave_lds = lds + (threadIdx.x >> 6) * tn::KP_WAVE_LDS;
    const int64_t n_tiles = (n + 31) >> 5;
    const int C = a.C, FD = a.n_scales * C;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const int64_t rr = valid ? row : 0;
        const float xs[3] = {x[rr * x_stride], x[rr * x_stride + 1], x[rr * x_stride + 2]};
        for (int s = 0; s < a.n_scales; ++s) {
            f32x4k g[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q)
                g[q] = valid ? reinterpret_cast<const f32x4k *>(grad_feat + row * FD + s * C + h * (C / 2))[q]
                             : f32x4k{0.f, 0.f, 0.f, 0.f};
            const float *const pl[3] = {a.planes[s][0], a.planes[s][1], a.planes[s][2]};
            float *const gr[3] = {a.grads[s][0], a.grads[s][1], a.grads[s][2]};
            tn::kp_scatter_scale<NV, 4>(pl, gr, a.H[s], a.W[s], C, xs, valid, g, h * (C / 2), wave_lds, j, h);
        }
    }
}

int make_args(const tn_kplanes_desc *d, KpArgs &a, float *const (*grads)[3])
{
    TN_REQUIRE(d, TN_E_NULL, "kplanes: null descriptor");
    TN_REQUIRE(d->n_scales >= 1 && d->n_scales <= TN_KPLANES_MAX_SCALES, TN_E_CONFIG, "kplanes: n_scales out of range");
    TN_REQUIRE(d->channels == 8 || d->channels == 16 || d->channels == 32, TN_E_CONFIG, "kplanes: channels must be 8, 16 or 32");
    a.n_scales = d->n_scales; a.C = d->channels;
    for (int s = 0; s < d->n_scales; ++s) {
        TN_REQUIRE(d->height[s] > 0 && d->width[s] > 0, TN_E_SIZE, "kplanes: bad plane resolution");
        TN_REQUIRE((int64_t)d->height[s] * d->width[s] * d->channels < (1ll << 30), TN_E_SIZE, "kplanes: plane too large (2^30 elements)");
        a.H[s] = d->height[s]; a.W[s] = d->width[s];
        TN_REQUIRE(d->planes[s][0], TN_E_NULL, "kplanes: null plane pointer");
        for (int p = 0; p < 3; ++p) {     // planes 1 and 2 may be NULL: the factor is then 1 (single-plane lookup)
            TN_REQUIRE(((uintptr_t)d->planes[s][p] & 15) == 0, TN_E_ALIGN, "kplanes: planes must be 16-byte aligned");
            a.planes[s][p] = d->planes[s][p];
            a.grads[s][p] = nullptr;
            if (grads && d->planes[s][p]) {
                TN_REQUIRE(grads[s][p], TN_E_NULL, "kplanes: null grad plane pointer");
                a.grads[s][p] = grads[s][p];
            }
        }
    }
    return TN_OK;
}

// ---- explicit K-Planes decoders (models.py:183-205): out[n,k] = act(sum_c f[n,c] * basis[n,k,c]) --------------------------
// A wave walks samples; lanes stride over the C channels (coalesced 256-B runs of f and of every basis row), the K dot
// products are reduced across the wave.  HBM-bound: (1 + K) * C * 4 B per sample forward, twice that backward.
template <int ACT>
__device__ __forceinline__ float bd_act(float v)
{
    if (ACT == TN_ACT_EXP_M1) return expf(v - 1.0f);
    if (ACT == TN_ACT_EXP) return expf(v);
    if (ACT == TN_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int ACT>
__global__ __launch_bounds__(256) void basis_dot_fwd_kernel(const float *__restrict__ f, const float *__restrict__ basis, int64_t n, int C,
                                                            int K, float *__restrict__ out)
{
    const int lane = tn::lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += n_waves) {
        const float *fi = f + i * C, *bi = basis + i * (int64_t)K * C;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int c = lane; c < C; c += 64) acc += fi[c] * bi[k * C + c];
            acc = tn::wave_sum(acc);
            if (lane == 0) out[i * K + k] = bd_act<ACT>(acc);
        }
    }
}

// g_basis[n,k,c] = gv[n,k] f[n,c];  g_f[n,c] (+)= sum_k gv[n,k] basis[n,k,c];  gv = g * act'(v), with the truncated
// exponential's clamp (models.py:50-53: exp(clamp(x, -15, 15)), x = v - 1) and sigmoid' = y (1 - y)
template <int ACT>
__global__ __launch_bounds__(256) void basis_dot_bwd_kernel(const float *__restrict__ f, const float *__restrict__ basis,
                                                            const float *__restrict__ g_out, int64_t n, int C, int K,
                                                            float *__restrict__ g_f, float *__restrict__ g_basis, int accumulate_f)
{
    const int lane = tn::lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += n_waves) {
        const float *fi = f + i * C, *bi = basis + i * (int64_t)K * C;
        float gv[4];
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int c = lane; c < C; c += 64) acc += fi[c] * bi[k * C + c];
            acc = tn::wave_sum(acc);
            const float g = g_out[i * K + k];
            if (ACT == TN_ACT_EXP_M1) gv[k] = g * expf(fminf(fmaxf(acc - 1.0f, -15.0f), 15.0f));
            else if (ACT == TN_ACT_EXP) gv[k] = g * expf(fminf(fmaxf(acc, -15.0f), 15.0f));
            else if (ACT == TN_ACT_SIGMOID) { const float y = 1.0f / (1.0f + expf(-acc)); gv[k] = g * y * (1.0f - y); }
            else gv[k] = g;
        }
        for (int c = lane; c < C; c += 64) {
            const float fc = fi[c];
            float a = accumulate_f ? g_f[i * C + c] : 0.0f;
            for (int k = 0; k < K; ++k) {
                a += gv[k] * bi[k * C + c];
                if (g_basis) g_basis[(i * K + k) * C + c] = gv[k] * fc;
            }
            g_f[i * C + c] = a;
        }
    }
}

inline unsigned tile_blocks(int64_t n) { return (unsigned)std::min<int64_t>(((n + 31) / 32 + 3) / 4, 256 * 8); }

}  // namespace

extern "C" int tn_kplanes_fwd(const tn_kplanes_desc *desc, const float *x, int64_t x_stride, int64_t n, float *feat,
                              void *stream)
{
    KpArgs a;
    if (int rc = make_args(desc, a, nullptr)) return rc;
    TN_REQUIRE(n >= 0 && x_stride >= 3, TN_E_SIZE, "tn_kplanes_fwd: bad size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && feat, TN_E_NULL, "tn_kplanes_fwd: null pointer");
    TN_REQUIRE(((uintptr_t)feat & 15) == 0, TN_E_ALIGN, "tn_kplanes_fwd: feat must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(tile_blocks(n)), block(256);
    switch (a.C) {
    case 8: kplanes_fwd_kernel<1><<<grid, block, 0, s>>>(a, x, x_stride, n, feat); break;
    case 16: kplanes_fwd_kernel<2><<<grid, block, 0, s>>>(a, x, x_stride, n, feat); break;
    default: kplanes_fwd_kernel<4><<<grid, block, 0, s>>>(a, x, x_stride, n, feat); break;
    }
    return tn::check_launch("kplanes_fwd_kernel");
}

extern "C" int tn_kplanes_bwd(const tn_kplanes_desc *desc, const float *x, int64_t x_stride, int64_t n,
                              const float *grad_feat, float *const (*grad_planes)[3], void *stream)
{
    KpArgs a;
    TN_REQUIRE(grad_planes, TN_E_NULL, "tn_kplanes_bwd: null grad_planes");
    if (int rc = make_args(desc, a, grad_planes)) return rc;
    TN_REQUIRE(n >= 0 && x_stride >= 3, TN_E_SIZE, "tn_kplanes_bwd: bad size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && grad_feat, TN_E_NULL, "tn_kplanes_bwd: null pointer");
    TN_REQUIRE(((uintptr_t)grad_feat & 15) == 0, TN_E_ALIGN, "tn_kplanes_bwd: grad_feat must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(tile_blocks(n)), block(256);
    switch (a.C) {
    case 8: kplanes_bwd_kernel<1><<<grid, block, 0, s>>>(a, x, x_stride, n, grad_feat); break;
    case 16: kplanes_bwd_kernel<2><<<grid, block, 0, s>>>(a, x, x_stride, n, grad_feat); break;
    default: kplanes_bwd_kernel<4><<<grid, block, 0, s>>>(a, x, x_stride, n, grad_feat); break;
    }
    return tn::check_launch("kplanes_bwd_kernel");
}

extern "C" int tn_basis_dot_fwd(const float *f, const float *basis, int64_t n, int32_t channels, int32_t n_out, int32_t activation,
                                float *out, void *stream)
{
    TN_REQUIRE(n >= 0 && channels > 0 && n_out >= 1 && n_out <= 4, TN_E_SIZE, "tn_basis_dot_fwd: bad size (1 <= n_out <= 4)");
    if (n == 0) return TN_OK;
    TN_REQUIRE(f && basis && out, TN_E_NULL, "tn_basis_dot_fwd: null pointer");
    const dim3 grid((unsigned)std::min<int64_t>((n + 3) / 4, 256 * 8)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (activation) {
    case TN_ACT_EXP_M1: basis_dot_fwd_kernel<TN_ACT_EXP_M1><<<grid, block, 0, s>>>(f, basis, n, channels, n_out, out); break;
    case TN_ACT_SIGMOID: basis_dot_fwd_kernel<TN_ACT_SIGMOID><<<grid, block, 0, s>>>(f, basis, n, channels, n_out, out); break;
    case TN_ACT_EXP: basis_dot_fwd_kernel<TN_ACT_EXP><<<grid, block, 0, s>>>(f, basis, n, channels, n_out, out); break;
    case TN_ACT_NONE: basis_dot_fwd_kernel<TN_ACT_NONE><<<grid, block, 0, s>>>(f, basis, n, channels, n_out, out); break;
    default: return tn::fail(TN_E_CONFIG, "tn_basis_dot_fwd: unknown activation");
    }
    return tn::check_launch("basis_dot_fwd_kernel");
}

extern "C" int tn_basis_dot_bwd(const float *f, const float *basis, const float *grad_out, int64_t n, int32_t channels, int32_t n_out,
                                int32_t activation, float *grad_f, float *grad_basis, int32_t accumulate_f, void *stream)
{
    TN_REQUIRE(n >= 0 && channels > 0 && n_out >= 1 && n_out <= 4, TN_E_SIZE, "tn_basis_dot_bwd: bad size (1 <= n_out <= 4)");
    if (n == 0) return TN_OK;
    TN_REQUIRE(f && basis && grad_out && grad_f, TN_E_NULL, "tn_basis_dot_bwd: null pointer");
    const dim3 grid((unsigned)std::min<int64_t>((n + 3) / 4, 256 * 8)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (activation) {
    case TN_ACT_EXP_M1: basis_dot_bwd_kernel<TN_ACT_EXP_M1><<<grid, block, 0, s>>>(f, basis, grad_out, n, channels, n_out, grad_f, grad_basis, accumulate_f); break;
    case TN_ACT_SIGMOID: basis_dot_bwd_kernel<TN_ACT_SIGMOID><<<grid, block, 0, s>>>(f, basis, grad_out, n, channels, n_out, grad_f, grad_basis, accumulate_f); break;
    case TN_ACT_EXP: basis_dot_bwd_kernel<TN_ACT_EXP><<<grid, block, 0, s>>>(f, basis, grad_out, n, channels, n_out, grad_f, grad_basis, accumulate_f); break;
    case TN_ACT_NONE: basis_dot_bwd_kernel<TN_ACT_NONE><<<grid, block, 0, s>>>(f, basis, grad_out, n, channels, n_out, grad_f, grad_basis, accumulate_f); break;
    default: return tn::fail(TN_E_CONFIG, "tn_basis_dot_bwd: unknown activation");
    }
    return tn::check_launch("basis_dot_bwd_kernel");
}
