// K-Planes feature field (reference src/models.py:93-163): 3 scales x 3 planes of bilinear
// lookups, Hadamard product over the planes of a scale, scales concatenated.
// Replaces 9 x (grid_sampler_2d + transpose + contiguous) + 6 muls + cat of the reference
// (27+ launches) with one launch; backward scatters with hardware fp32 atomics.
#include "kplanes_device.h"
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

// ---- backward: transposed, run-merged scatter ------------------------------------------------------
// Measured on MI355X (scratch/atomic_bench.hip): a wave64 global_atomic_add_f32 whose lanes hit 64 different
// cache lines retires ~20 G lane-atomics/s, one whose half-waves each cover the 32 consecutive dwords of ONE
// line ~270 G/s.  The forward mapping (lane = sample) is the slow pattern, so the scatter is transposed
// through a 5 KiB per-wave LDS tile: phase A (lane = sample, channel half) writes the 32x32 tile of
// d(feat)/d(plane value) plus the 4 tap offsets / weights of every sample; phase B (lane = channel) walks the
// samples in order -- consecutive samples of a ray fall into the same cell for several steps (a straight
// line visits the cells of a plane monotonically), so the contributions of a RUN of samples are accumulated
// in registers and ONE full-line atomic per tap is issued at the end of the run.  Run boundaries depend on
// the cell only, so they are the same for all four taps: phase A ballots them into a 32-bit scalar mask and
// phase B's control flow is scalar (s_bitcmp + s_cbranch, no exec-mask divergence).  Half-wave 0 handles the
// taps (nw, ne), half-wave 1 (sw, se).
constexpr int GS = 36;                                   // floats per tile row: conflict-free b128 writes
constexpr int KP_WAVE_LDS = 32 * GS + 2 * 4 * 32;        // tile + offsets + weights (floats)

template <int NV>
__global__ __launch_bounds__(256, 3) void kplanes_bwd_kernel(KpArgs a, const float *__restrict__ x, int64_t x_stride,
                                                          int64_t n, const float *__restrict__ grad_feat)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * KP_WAVE_LDS];
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    float *tileG = lds + (threadIdx.x >> 6) * KP_WAVE_LDS;
    int *tileO = reinterpret_cast<int *>(tileG + 32 * GS);
    float *tileW = tileG + 32 * GS + 4 * 32;
    const int64_t n_tiles = (n + 31) >> 5;
    const int C = a.C, FD = a.n_scales * C;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const int64_t rr = valid ? row : 0;
        const float xs[3] = {x[rr * x_stride], x[rr * x_stride + 1], x[rr * x_stride + 2]};
        for (int s = 0; s < a.n_scales; ++s) {
            tn::PlaneTaps t[3];
            f32x4k val[3][NV];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float u, v;
                tn::pair_uv(xs, p, u, v);
                t[p] = tn::plane_taps(u, v, a.H[s], a.W[s], C);
                if (a.planes[s][p]) {
                    tn::plane_gather<NV>(a.planes[s][p], t[p], h * (C / 2), val[p]);
                    __builtin_amdgcn_sched_barrier(0);      // one plane's 16 loads in flight at a time (register budget)
                } else {
#pragma unroll
                    for (int q = 0; q < NV; ++q) val[p][q] = f32x4k{1.f, 1.f, 1.f, 1.f};
                }
            }
            f32x4k g[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q)
                g[q] = valid ? reinterpret_cast<const f32x4k *>(grad_feat + row * FD + s * C + h * (C / 2))[q]
                             : f32x4k{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (a.grads[s][p] == nullptr) continue;
                // ---- phase A: lane = (sample j, channel half h) ----
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const f32x4k gp = p == 0 ? g[q] * val[1][q] * val[2][q]
                                             : (p == 1 ? g[q] * val[0][q] * val[2][q] : g[q] * (val[0][q] * val[1][q]));
                    *reinterpret_cast<f32x4k *>(tileG + j * GS + h * (C / 2) + 4 * q) = gp;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {          // this lane publishes taps 2h, 2h+1 of its sample
                    const int o0 = t[p].off[0 + k], o1 = t[p].off[2 + k];
                    const float w0 = t[p].w[0 + k], w1 = t[p].w[2 + k];
                    tileO[(2 * h + k) * 32 + j] = valid ? (h ? o1 : o0) : -1;
                    tileW[(2 * h + k) * 32 + j] = h ? w1 : w0;
                }
                // run boundaries: sample j closes a run when the next sample falls into another cell.  When the next
                // cell is a 4-neighbour, two of the four texels are shared with it: instead of flushing them, their
                // partial sums are carried into the next run (x moves: within the half-wave; y moves: across halves).
                const int cell = valid ? t[p].cell : -1 - j;
                const int next_cell = __shfl_down(cell, 1, 64);
                const int dcell = (j < 31) ? next_cell - cell : 0x40000000;
                const int rowlen = a.W[s] + 4;
                const unsigned run_end = (unsigned)__ballot(dcell != 0);                       // low 32 bits: half 0 == half 1
                const unsigned mv_xp = (unsigned)__ballot(dcell == 1), mv_xm = (unsigned)__ballot(dcell == -1);
                const unsigned mv_yp = (unsigned)__ballot(dcell == rowlen), mv_ym = (unsigned)__ballot(dcell == -rowlen);
                asm volatile("" ::: "memory");         // DS ops of one wave execute in order; only the compiler must not reorder
                // ---- phase B: lane = (tap pair h, channel c) ----
                const int c = j;                       // channel
                float *gbase = a.grads[s][p] + c;
                if (c < C) {
                    const int *O0 = tileO + (2 * h) * 32, *O1 = O0 + 32;
                    const f32x4k *W0 = reinterpret_cast<const f32x4k *>(tileW + (2 * h) * 32);
                    const f32x4k *W1 = W0 + 8;
                    float a0 = 0.0f, a1 = 0.0f;        // running sums of this half's left / right texel
                    // (not unrolled: fully unrolled, the 32-sample walk with its five-way run logic made the kernel 12 k
                    // instructions = 72 KB, more than the instruction cache two CUs share)
#pragma clang loop unroll(disable)
                    for (int s4 = 0; s4 < 8; ++s4) {
                        const f32x4k w0 = W0[s4], w1 = W1[s4];
                        float gv[4];                   // four samples at a time: 32 at once cost 12 spilled registers
#pragma unroll
                        for (int u = 0; u < 4; ++u) gv[u] = tileG[(4 * s4 + u) * GS + c];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sI = 4 * s4 + u;
                            a0 = fmaf(gv[u], w0[u], a0);
                            a1 = fmaf(gv[u], w1[u], a1);
                            if ((run_end >> sI) & 1u) {          // wave-uniform (scalar) control flow from here on
                                const int o0 = O0[sI], o1 = O1[sI];
                                if ((mv_xp >> sI) & 1u) {        // next cell = x+1: right texel becomes the left one
                                    if (o0 >= 0) atomicAdd(gbase + o0, a0);
                                    a0 = a1; a1 = 0.0f;
                                } else if ((mv_xm >> sI) & 1u) { // next cell = x-1
                                    if (o1 >= 0) atomicAdd(gbase + o1, a1);
                                    a1 = a0; a0 = 0.0f;
                                } else if ((mv_yp >> sI) & 1u) { // next cell = y+1: the lower row (half 1) becomes the upper row
                                    const float t0 = __shfl_xor(a0, 32, 64), t1 = __shfl_xor(a1, 32, 64);
                                    if (h == 0) {
                                        if (o0 >= 0) atomicAdd(gbase + o0, a0);
                                        if (o1 >= 0) atomicAdd(gbase + o1, a1);
                                    }
                                    a0 = h == 0 ? t0 : 0.0f; a1 = h == 0 ? t1 : 0.0f;
                                } else if ((mv_ym >> sI) & 1u) { // next cell = y-1
                                    const float t0 = __shfl_xor(a0, 32, 64), t1 = __shfl_xor(a1, 32, 64);
                                    if (h == 1) {
                                        if (o0 >= 0) atomicAdd(gbase + o0, a0);
                                        if (o1 >= 0) atomicAdd(gbase + o1, a1);
                                    }
                                    a0 = h == 1 ? t0 : 0.0f; a1 = h == 1 ? t1 : 0.0f;
                                } else {
                                    if (o0 >= 0) atomicAdd(gbase + o0, a0);
                                    if (o1 >= 0) atomicAdd(gbase + o1, a1);
                                    a0 = 0.0f; a1 = 0.0f;
                                }
                            }
                        }
                    }
                }
                asm volatile("" ::: "memory");
            }
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
        TN_REQUIRE((int64_t)d->height[s] * d->width[s] * d->channels < (1ll << 31), TN_E_SIZE, "kplanes: plane too large");
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
    case TN_ACT_NONE: basis_dot_bwd_kernel<TN_ACT_NONE><<<grid, block, 0, s>>>(f, basis, grad_out, n, channels, n_out, grad_f, grad_basis, accumulate_f); break;
    default: return tn::fail(TN_E_CONFIG, "tn_basis_dot_bwd: unknown activation");
    }
    return tn::check_launch("basis_dot_bwd_kernel");
}
