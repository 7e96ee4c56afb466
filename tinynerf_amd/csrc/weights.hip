// NeRF volume-rendering weights and per-ray compositing for packed variable-length rays.
//
// Replaces reference src/cuda.cu:3-58 (one *thread* per ray, serial loop, swapped launch
// dims) with one *wavefront* per ray: the 64 lanes load 64 consecutive samples of the ray
// coalesced, alpha = exp(-sigma*delta) is evaluated lane-parallel, and the transmittance is a
// wave-wide multiplicative exclusive scan (6 DPP/shuffle steps) carried across 64-sample
// chunks.  HBM-bound: fwd 12 B/sample (+8 B/ray), bwd 20 B/sample (+8 B/ray).
//
// Semantics kept from the reference:
//  fwd (cuda.cu:19-28): w_k = T_k (1-alpha_k) while T_k > threshold, T_k = prod_{j<k} alpha_j;
//      all samples from the first k with T_k <= threshold on get 0.  The product T*(1-alpha) is
//      formed in fp64 and rounded once, as the reference's double literals do (cuda.cu:25).
//  bwd (cuda.cu:49-56): gs_k = delta_k (T_{k+1} g_k - sum_{j>k} w_j g_j), NO termination.
#include "tn_common.h"
#include <algorithm>

namespace {

constexpr int WAVES_PER_BLOCK = 4;

// inclusive multiplicative scan across the wave
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// COMP: the ray's composite (core.py:256-265, composite_fwd_kernel below) in the same pass -- the same per-lane partial sums
// in the same order, so `rendered` is bit-identical to the two-launch form.
template <bool COMP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void weights_fwd_kernel(
    const float *__restrict__ sigmas, const float *__restrict__ steps, const int32_t *__restrict__ info,
    float threshold, float *__restrict__ weights, int64_t n_rays, float *__restrict__ gate,
    const float *__restrict__ rgbs = nullptr, const float *__restrict__ bg = nullptr, float *__restrict__ rendered = nullptr)
{
    float cr = 0.f, cg = 0.f, cb = 0.f, co = 0.f;
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int2 sc = reinterpret_cast<const int2 *>(info)[ray];
    const int start = sc.x, count = sc.y;
    float carry = 1.0f;          // transmittance entering the chunk
    bool alive = true;           // wave-uniform: no lane has terminated yet
    bool positive = false;       // this lane has written a weight > 0
    for (int base = 0; base < count; base += 64) {
        const int k = base + lane;
        const bool valid = k < count;
        float w = 0.0f;
        if (alive) {
            float a = 1.0f;
            if (valid) a = expf(-sigmas[start + k] * steps[start + k]);
            const float incl = wave_scan_mul(a, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = carry * excl;
            // the reference's while-loop stops at the FIRST k with !(T_k > threshold)
            const uint64_t dead = __ballot(valid && !(T > threshold));
            const int first_dead = dead ? __builtin_ctzll(dead) : 64;
            if (lane < first_dead) w = (float)((double)T * (1.0 - (double)a));
            carry = carry * __shfl(incl, 63, 64);
            alive = dead == 0;
        }
        if (valid) weights[start + k] = w;
        positive = positive || w > 0.0f;
        if constexpr (COMP) {
            if (valid) {
                if (w != 0.0f) {     // masked samples carry rgb = 0 in the reference (core.py:248-250)
                    const float *c = rgbs + 3 * (int64_t)(start + k);
                    cr += c[0] * w; cg += c[1] * w; cb += c[2] * w;
                }
                co += w;
            }
        }
    }
    if constexpr (COMP) {
        cr = tn::wave_sum(cr); cg = tn::wave_sum(cg); cb = tn::wave_sum(cb); co = tn::wave_sum(co);
        if (lane == 0) {
            if (bg) { cr += bg[0] * (1.f - co); cg += bg[1] * (1.f - co); cb += bg[2] * (1.f - co); }
            rendered[3 * ray + 0] = cr; rendered[3 * ray + 1] = cg; rendered[3 * ray + 2] = cb;
        }
    }
    // "Empty iteration" flag of the harness (core.py:251-254): raised by every ray that has a weight > 0.  All writers store the
    // same value, so plain stores do (no atomic: 22 000 waves on one address); once it is up the rest only read it.
    if (gate != nullptr && __ballot(positive) != 0 && lane == 0 && gate[0] == 0.0f) gate[0] = 1.0f;
}

// Single pass over HBM for rays of up to 64*MAXC samples: w*g and alpha stay in registers
// between the reduction (pass 1 of cuda.cu:51) and the prefix sweep (pass 2, cuda.cu:52-56).
// COMP: d loss / d weights is not read but formed here from the composite's upstream gradient (composite_bwd_kernel below: <rgb, g>
// - <bg, g>, and grad_rgbs = w g written on the way) -- the same arithmetic, one launch and no grad_weights round trip.
template <int MAXC, bool COMP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void weights_bwd_kernel(
    const float *__restrict__ sigmas, const float *__restrict__ steps, const int32_t *__restrict__ info,
    const float *__restrict__ weights, const float *__restrict__ grad_w, float *__restrict__ grad_sigmas,
    int64_t n_rays, const float *__restrict__ rgbs = nullptr, const float *__restrict__ bg = nullptr,
    const float *__restrict__ grad_rendered = nullptr, float *__restrict__ grad_rgbs = nullptr)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int2 sc = reinterpret_cast<const int2 *>(info)[ray];
    const int start = sc.x, count = sc.y;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gbg = 0.f;
    if constexpr (COMP) {
        g0 = grad_rendered[3 * ray]; g1 = grad_rendered[3 * ray + 1]; g2 = grad_rendered[3 * ray + 2];
        gbg = bg ? (bg[0] * g0 + bg[1] * g1 + bg[2] * g2) : 0.f;
    }
    // upstream gradient of sample i's weight (w = its weight)
    auto grad_weight = [&](int64_t i, float w) -> float {
        if constexpr (COMP) {
            grad_rgbs[3 * i] = w * g0; grad_rgbs[3 * i + 1] = w * g1; grad_rgbs[3 * i + 2] = w * g2;
            float d = 0.f;
            if (w != 0.0f) d = rgbs[3 * i] * g0 + rgbs[3 * i + 1] * g1 + rgbs[3 * i + 2] * g2;
            return d - gbg;
        } else {
            return grad_w[i];
        }
    };
    if (count <= 64 * MAXC) {
        float wg[MAXC], al[MAXC], dl[MAXC], gg[MAXC];
        float total = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = c * 64 + lane;
            wg[c] = 0.f; al[c] = 1.f; dl[c] = 0.f; gg[c] = 0.f;
            if (c * 64 < count && k < count) {
                const float w = weights[start + k];
                const float g = grad_weight(start + k, w);
                dl[c] = steps[start + k];
                al[c] = expf(-sigmas[start + k] * dl[c]);
                wg[c] = w * g;
                gg[c] = g;
            }
            total += wg[c];
        }
        total = tn::wave_sum(total);
        float acc = -total, T = 1.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c * 64 < count) {
                const int k = c * 64 + lane;
                const float ps = wave_scan_add(wg[c], lane);
                const float pt = wave_scan_mul(al[c], lane);
                if (k < count) grad_sigmas[start + k] = dl[c] * ((acc + ps) + (T * pt) * gg[c]);
                acc += __shfl(ps, 63, 64);
                T *= __shfl(pt, 63, 64);
            }
        }
    } else {
        float total = 0.0f;
        for (int k = lane; k < count; k += 64) { const float w = weights[start + k]; total += w * grad_weight(start + k, w); }
        total = tn::wave_sum(total);
        float acc = -total, T = 1.0f;
        for (int base = 0; base < count; base += 64) {
            const int k = base + lane;
            float wgk = 0.f, a = 1.f, d = 0.f, g = 0.f;
            if (k < count) {
                const float w = weights[start + k];
                g = grad_weight(start + k, w);
                d = steps[start + k];
                a = expf(-sigmas[start + k] * d);
                wgk = w * g;
            }
            const float ps = wave_scan_add(wgk, lane);
            const float pt = wave_scan_mul(a, lane);
            if (k < count) grad_sigmas[start + k] = d * ((acc + ps) + (T * pt) * g);
            acc += __shfl(ps, 63, 64);
            T *= __shfl(pt, 63, 64);
        }
    }
}

// reference core.py:256-265 -- per-ray segmented sum, deterministic (fixed tree order).
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void composite_fwd_kernel(
    const float *__restrict__ rgbs, const float *__restrict__ weights, const int32_t *__restrict__ info,
    const float *__restrict__ bg, float *__restrict__ rendered, float *__restrict__ opacity, int64_t n_rays)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int2 sc = reinterpret_cast<const int2 *>(info)[ray];
    float r = 0.f, g = 0.f, b = 0.f, o = 0.f;
    for (int k = lane; k < sc.y; k += 64) {
        const float w = weights[sc.x + k];
        if (w != 0.0f) {     // masked samples carry rgb = 0 in the reference (core.py:248-250)
            const float *c = rgbs + 3 * (int64_t)(sc.x + k);
            r += c[0] * w; g += c[1] * w; b += c[2] * w;
        }
        o += w;
    }
    r = tn::wave_sum(r); g = tn::wave_sum(g); b = tn::wave_sum(b); o = tn::wave_sum(o);
    if (lane == 0) {
        if (bg) { r += bg[0] * (1.f - o); g += bg[1] * (1.f - o); b += bg[2] * (1.f - o); }
        rendered[3 * ray + 0] = r; rendered[3 * ray + 1] = g; rendered[3 * ray + 2] = b;
        if (opacity) opacity[ray] = o;
    }
}

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void composite_bwd_kernel(
    const float *__restrict__ rgbs, const float *__restrict__ weights, const int32_t *__restrict__ info,
    const float *__restrict__ bg, const float *__restrict__ grad_rendered, float *__restrict__ grad_rgbs,
    float *__restrict__ grad_weights, int64_t n_rays)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int2 sc = reinterpret_cast<const int2 *>(info)[ray];
    const float g0 = grad_rendered[3 * ray], g1 = grad_rendered[3 * ray + 1], g2 = grad_rendered[3 * ray + 2];
    const float gbg = bg ? (bg[0] * g0 + bg[1] * g1 + bg[2] * g2) : 0.f;
    for (int k = lane; k < sc.y; k += 64) {
        const int64_t i = sc.x + k;
        const float w = weights[i];
        if (grad_rgbs) { grad_rgbs[3 * i] = w * g0; grad_rgbs[3 * i + 1] = w * g1; grad_rgbs[3 * i + 2] = w * g2; }
        if (grad_weights) {
            float d = 0.f;
            if (w != 0.0f) d = rgbs[3 * i] * g0 + rgbs[3 * i + 1] * g1 + rgbs[3 * i + 2] * g2;
            grad_weights[i] = d - gbg;
        }
    }
}

// d/d rendered of  c * sum (rendered - target)^2  and the sum itself (fp64 accumulator), one pass (run.py:252,259)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float *__restrict__ r, const float *__restrict__ t, int64_t n, float cg,
                                                       const float *__restrict__ cg_dev, float *__restrict__ grad, double *__restrict__ sumsq,
                                                       const float *__restrict__ gate)
{
    float c = cg_dev ? cg * cg_dev[0] : cg;
    if (gate != nullptr && !(gate[0] > 0.0f)) c = 0.0f;     // "Empty iteration": the image loss reaches no parameter
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = r[i] - t[i];
        grad[i] = d * c;
        s = fmaf(d, d, s);
    }
    double ds = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ds;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sumsq, red[0] + red[1] + red[2] + red[3]);
}

// per-sample ray index, contiguous step sizes and per-ray direction out of (packed, info): what the per-ray colour-head
// table and the weights kernels index (one launch instead of repeat_interleave + gathers + copies)
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void ray_aux_kernel(const float *__restrict__ packed, const int32_t *__restrict__ info,
                                                                       int64_t n_rays, int32_t *__restrict__ ray_ids,
                                                                       float *__restrict__ steps, float *__restrict__ dirs)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int2 sc = reinterpret_cast<const int2 *>(info)[ray];
    if (lane < 3) dirs[3 * ray + lane] = sc.y > 0 ? packed[7 * (int64_t)sc.x + 3 + lane] : 0.0f;
    for (int k = lane; k < sc.y; k += 64) {
        const int64_t i = sc.x + k;
        ray_ids[i] = (int32_t)ray;
        steps[i] = packed[7 * i + 6];
    }
}

inline unsigned ray_blocks(int64_t n_rays) { return (unsigned)((n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

}  // namespace

extern "C" int tn_weights_fwd(const float *sigmas, const float *steps, const int32_t *info, float threshold,
                              float *weights, int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_weights_fwd: negative size");
    if (n_rays == 0 || n_samples == 0) return TN_OK;
    TN_REQUIRE(sigmas && steps && info && weights, TN_E_NULL, "tn_weights_fwd: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_weights_fwd: info must be 8-byte aligned");
    weights_fwd_kernel<false><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(sigmas, steps, info, threshold,
                                                                                                                 weights, n_rays, nullptr);
    return tn::check_launch("weights_fwd_kernel");
}

extern "C" int tn_weights_fwd_gate(const float *sigmas, const float *steps, const int32_t *info, float threshold, float *weights,
                                   float *gate, int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_weights_fwd_gate: negative size");
    if (n_rays == 0 || n_samples == 0) return TN_OK;
    TN_REQUIRE(sigmas && steps && info && weights && gate, TN_E_NULL, "tn_weights_fwd_gate: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_weights_fwd_gate: info must be 8-byte aligned");
    weights_fwd_kernel<false><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(sigmas, steps, info, threshold,
                                                                                                                 weights, n_rays, gate);
    return tn::check_launch("weights_fwd_kernel");
}

extern "C" int tn_render_rays_fwd(const float *sigmas, const float *steps, const float *rgbs, const int32_t *info, const float *bg,
                                  float threshold, float *weights, float *rendered, float *gate, int64_t n_samples, int64_t n_rays,
                                  void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_render_rays_fwd: negative size");
    if (n_rays == 0) return TN_OK;
    TN_REQUIRE(info && weights && rendered && (n_samples == 0 || (sigmas && steps && rgbs)), TN_E_NULL, "tn_render_rays_fwd: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_render_rays_fwd: info must be 8-byte aligned");
    weights_fwd_kernel<true><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(sigmas, steps, info, threshold,
                                                                                                                weights, n_rays, gate, rgbs, bg, rendered);
    return tn::check_launch("weights_fwd_kernel(composite)");
}

extern "C" int tn_render_rays_bwd(const float *sigmas, const float *steps, const float *rgbs, const int32_t *info, const float *bg,
                                  const float *weights, const float *grad_rendered, float *grad_rgbs, float *grad_sigmas,
                                  int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_render_rays_bwd: negative size");
    if (n_rays == 0 || n_samples == 0) return TN_OK;
    TN_REQUIRE(sigmas && steps && rgbs && info && weights && grad_rendered && grad_rgbs && grad_sigmas, TN_E_NULL, "tn_render_rays_bwd: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_render_rays_bwd: info must be 8-byte aligned");
    weights_bwd_kernel<16, true><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(
        sigmas, steps, info, weights, nullptr, grad_sigmas, n_rays, rgbs, bg, grad_rendered, grad_rgbs);
    return tn::check_launch("weights_bwd_kernel(composite)");
}

extern "C" int tn_weights_bwd(const float *sigmas, const float *steps, const int32_t *info, const float *weights,
                              const float *grad_weights, float *grad_sigmas, int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_weights_bwd: negative size");
    if (n_rays == 0 || n_samples == 0) return TN_OK;
    TN_REQUIRE(sigmas && steps && info && weights && grad_weights && grad_sigmas, TN_E_NULL, "tn_weights_bwd: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_weights_bwd: info must be 8-byte aligned");
    weights_bwd_kernel<16, false><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(sigmas, steps, info, weights,
                                                                                                                     grad_weights, grad_sigmas, n_rays);
    return tn::check_launch("weights_bwd_kernel");
}

extern "C" int tn_composite_fwd(const float *rgbs, const float *weights, const int32_t *info, const float *bg,
                                float *rendered, float *opacity, int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_composite_fwd: negative size");
    if (n_rays == 0) return TN_OK;
    TN_REQUIRE(info && rendered && (n_samples == 0 || (rgbs && weights)), TN_E_NULL, "tn_composite_fwd: null pointer");
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream,
                       rgbs, weights, info, bg, rendered, opacity, n_rays);
    return tn::check_launch("composite_fwd_kernel");
}

extern "C" int tn_composite_bwd(const float *rgbs, const float *weights, const int32_t *info, const float *bg,
                                const float *grad_rendered, float *grad_rgbs, float *grad_weights,
                                int64_t n_samples, int64_t n_rays, void *stream)
{
    TN_REQUIRE(n_samples >= 0 && n_rays >= 0, TN_E_SIZE, "tn_composite_bwd: negative size");
    if (n_rays == 0 || n_samples == 0) return TN_OK;
    TN_REQUIRE(rgbs && weights && info && grad_rendered, TN_E_NULL, "tn_composite_bwd: null pointer");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream,
                       rgbs, weights, info, bg, grad_rendered, grad_rgbs, grad_weights, n_rays);
    return tn::check_launch("composite_bwd_kernel");
}

extern "C" int tn_mse_grad(const float *rendered, const float *target, int64_t n, float scale, const float *scale_dev, float *grad,
                           double *sumsq, void *stream)
{
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mse_grad: negative size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(rendered && target && grad && sumsq, TN_E_NULL, "tn_mse_grad: null pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 512);
    mse_grad_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(rendered, target, n, scale, scale_dev, grad, sumsq, nullptr);
    return tn::check_launch("mse_grad_kernel");
}

extern "C" int tn_mse_grad_gated(const float *rendered, const float *target, int64_t n, float scale, const float *scale_dev, const float *gate,
                                 float *grad, double *sumsq, void *stream)
{
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mse_grad_gated: negative size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(rendered && target && grad && sumsq && gate, TN_E_NULL, "tn_mse_grad_gated: null pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 512);
    mse_grad_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(rendered, target, n, scale, scale_dev, grad, sumsq, gate);
    return tn::check_launch("mse_grad_kernel");
}

// rows idx[i] of three [N, 3] ray tables in one launch (the harness' batch draw: origins, directions, target colours)
__global__ __launch_bounds__(256) void gather_rays_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c,
                                                          const int32_t *__restrict__ idx, int64_t n, float *__restrict__ oa,
                                                          float *__restrict__ ob, float *__restrict__ oc)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // element of the [n, 3] outputs
    if (e >= 3 * n) return;
    const int64_t r = e / 3;
    const int64_t src = (int64_t)idx[r] * 3 + (e - 3 * r);
    oa[e] = a[src];
    ob[e] = b[src];
    if (c != nullptr) oc[e] = c[src];
}

extern "C" int tn_gather_rays(const float *rays_o, const float *rays_d, const float *rgbs, const int32_t *idx, int64_t n, float *out_o,
                              float *out_d, float *out_rgb, void *stream)
{
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_gather_rays: negative size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(rays_o && rays_d && idx && out_o && out_d && (!rgbs || out_rgb), TN_E_NULL, "tn_gather_rays: null pointer");
    gather_rays_kernel<<<dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(rays_o, rays_d, rgbs, idx, n, out_o, out_d, out_rgb);
    return tn::check_launch("gather_rays_kernel");
}

extern "C" int tn_ray_aux(const float *packed, const int32_t *info, int64_t n_rays, int32_t *ray_ids, float *steps, float *dirs,
                          void *stream)
{
    TN_REQUIRE(n_rays >= 0, TN_E_SIZE, "tn_ray_aux: negative size");
    if (n_rays == 0) return TN_OK;
    TN_REQUIRE(packed && info && ray_ids && steps && dirs, TN_E_NULL, "tn_ray_aux: null pointer");
    TN_REQUIRE(((uintptr_t)info & 7) == 0, TN_E_ALIGN, "tn_ray_aux: info must be 8-byte aligned");
    ray_aux_kernel<<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(packed, info, n_rays, ray_ids, steps, dirs);
    return tn::check_launch("ray_aux_kernel");
}
