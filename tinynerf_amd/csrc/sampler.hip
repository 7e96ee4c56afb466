// Occupancy-grid ray sampling with variable-length sample packing.
//
// Replaces the ~25 ATen launches of reference src/core.py:165-188 (RayProvider) and the
// marchers / contractions / occupancy lookup it calls (core.py:11-88,147-156) with three
// launches that never materialise the [R,S,3] candidate tensors:
//   1. sample_mask : one wavefront per ray walks the S candidates 64 at a time, evaluates
//                    march -> jitter -> contraction -> trilinear occupancy test in registers and
//                    stores one 64-bit ballot per chunk plus the per-ray popcount;
//   2. sample_scan : int32 exclusive scan of the counts -> packing_info (start,count);
//   3. sample_pack : the same wavefront-per-ray walk; lanes whose mask bit is set recompute
//                    their sample (identical arithmetic -> identical bits) and write the 7-float
//                    row at start + rank, rank = popcount of lower mask bits (v_mbcnt).
//
// BIT-EXACTNESS.  packing_info and the packed coordinates must equal the reference's (torch
// ops, one IEEE rounding per op).  This translation unit is compiled with -ffp-contract=off and
// correctly-rounded fp32 division; every expression below follows the operation order of the
// cited reference line, and the trilinear lookup follows ATen's grid_sampler_3d (align_corners,
// zeros padding): unnormalise ((x+1)/2)*(size-1), floor, corner weights as products of
// differences, taps accumulated tnw,tne,tsw,tse,bnw,bne,bsw,bse with one rounded multiply and
// one rounded add each.
#include "tn_common.h"
#include <math.h>

namespace {

constexpr int WAVES_PER_BLOCK = 4;

struct SamplerArgs {
    int marcher, contraction, n_samples;
    int gd, gh, gw;
    float lo[3], hi[3];
    float near, far, step, threshold;
    const float *t_table, *delta_table, *grid, *jitter;
    uint64_t seed;
    int use_rng;
    const float *coarse;      // block maxima (4^3 voxels + 1 halo) or nullptr
    float inv_ext[3];         // pow2: 1 / (hi - lo) per axis, exact
    int pow2;                 // every box extent is a power of two: x / ext == x * inv_ext bit for bit (both are the correctly
};                            // rounded value of the same real number), and the 12-instruction IEEE division drops out

// ATen grid_sampler_3d forward for one point (bilinear, zeros padding, align_corners=True)
__device__ __forceinline__ float trilinear(const float *__restrict__ grid, int D, int H, int W,
                                           float x, float y, float z)
{
    const float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((z + 1.0f) / 2.0f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
    const float wx0 = x1 - ix, wx1 = ix - x0;
    const float wy0 = y1 - iy, wy1 = iy - y0;
    const float wz0 = z1 - iz, wz1 = iz - z0;
    // within-bounds tests on the float corner coordinates (NaN compares false -> tap skipped)
    const bool bx0 = x0 >= 0.0f && x0 < (float)W, bx1 = x1 >= 0.0f && x1 < (float)W;
    const bool by0 = y0 >= 0.0f && y0 < (float)H, by1 = y1 >= 0.0f && y1 < (float)H;
    const bool bz0 = z0 >= 0.0f && z0 < (float)D, bz1 = z1 >= 0.0f && z1 < (float)D;
    const int xi0 = bx0 ? (int)x0 : 0, xi1 = bx1 ? (int)x1 : 0;
    const int yi0 = by0 ? (int)y0 : 0, yi1 = by1 ? (int)y1 : 0;
    const int zi0 = bz0 ? (int)z0 : 0, zi1 = bz1 ? (int)z1 : 0;
    const int64_t r00 = ((int64_t)zi0 * H + yi0) * W, r01 = ((int64_t)zi0 * H + yi1) * W;
    const int64_t r10 = ((int64_t)zi1 * H + yi0) * W, r11 = ((int64_t)zi1 * H + yi1) * W;
    float acc = 0.0f;
    if (bx0 && by0 && bz0) acc = acc + grid[r00 + xi0] * ((wx0 * wy0) * wz0);
    if (bx1 && by0 && bz0) acc = acc + grid[r00 + xi1] * ((wx1 * wy0) * wz0);
    if (bx0 && by1 && bz0) acc = acc + grid[r01 + xi0] * ((wx0 * wy1) * wz0);
    if (bx1 && by1 && bz0) acc = acc + grid[r01 + xi1] * ((wx1 * wy1) * wz0);
    if (bx0 && by0 && bz1) acc = acc + grid[r10 + xi0] * ((wx0 * wy0) * wz1);
    if (bx1 && by0 && bz1) acc = acc + grid[r10 + xi1] * ((wx1 * wy0) * wz1);
    if (bx0 && by1 && bz1) acc = acc + grid[r11 + xi0] * ((wx0 * wy1) * wz1);
    if (bx1 && by1 && bz1) acc = acc + grid[r11 + xi1] * ((wx1 * wy1) * wz1);
    return acc;
}

// Conservative early reject: the interpolated value is a convex combination of the 8 taps (the rounded weights sum to
// 1 within a few ulp), so it cannot exceed the maximum of the block that covers every tap the point can read.  When
// that maximum is below 0.999 * threshold the candidate fails `value > threshold` without reading a tap; otherwise the
// exact lookup runs.  The mask is therefore identical, bit for bit, with and without the coarse grid.
__device__ __forceinline__ bool surely_empty(const SamplerArgs &a, float x, float y, float z)
{
    const float ix = ((x + 1.0f) / 2.0f) * (float)(a.gw - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(a.gh - 1);
    const float iz = ((z + 1.0f) / 2.0f) * (float)(a.gd - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    // all taps of an axis out of bounds (or NaN): the exact lookup returns 0, and 0 > threshold is false for threshold >= 0
    const bool inx = x0 >= -1.0f && x0 < (float)a.gw, iny = y0 >= -1.0f && y0 < (float)a.gh, inz = z0 >= -1.0f && z0 < (float)a.gd;
    if (!(inx && iny && inz)) return a.threshold >= 0.0f;
    const int bx = max((int)x0, 0) >> 2, by = max((int)y0, 0) >> 2, bz = max((int)z0, 0) >> 2;
    const int cw = (a.gw + 3) >> 2, ch = (a.gh + 3) >> 2;
    return a.coarse[((int64_t)bz * ch + by) * cw + bx] < a.threshold * 0.999f;
}

// per-ray constant of the AABB marcher: slab entry distance (reference core.py:78-81)
__device__ __forceinline__ float aabb_t_min(const SamplerArgs &a, const float o[3], const float d[3])
{
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float den = (d[c] == 0.0f) ? (d[c] + 1e-9f) : d[c];
        const float tl = (a.lo[c] - o[c]) / den;
        const float th = (a.hi[c] - o[c]) / den;
        m[c] = fminf(tl, th);
    }
    float t = fmaxf(fmaxf(m[0], m[1]), m[2]);
    t = fminf(fmaxf(t, a.near), a.far);
    return t;
}

template <int MARCH>
__device__ __forceinline__ void march_t(const SamplerArgs &a, float t_min, int k, float &t, float &delta)
{
    if constexpr (MARCH == TN_MARCH_AABB) {
        t = t_min + (float)k * a.step;                      // core.py:84-85
        delta = a.step;                                     // core.py:86
    } else {
        t = a.t_table[k];                                   // core.py:52-58
        delta = a.delta_table[k];
    }
}

// contraction of one world-space point; returns the in-box flag (always true for Mip-360)
template <int CONTRACT>
__device__ __forceinline__ bool contract(const SamplerArgs &a, const float p[3], float c[3])
{
    bool inside = true;
    if constexpr (CONTRACT == TN_CONTRACT_AABB) {           // core.py:29-30
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            inside = inside && (p[i] >= a.lo[i]) && (p[i] <= a.hi[i]);
            if (a.pow2) c[i] = ((p[i] - a.lo[i]) * a.inv_ext[i]) * 2.0f - 1.0f;     // (wave-uniform branch)
            else c[i] = (p[i] - a.lo[i]) / (a.hi[i] - a.lo[i]) * 2.0f - 1.0f;
        }
    } else {                                                // core.py:18-19
        float n;
        if constexpr (CONTRACT == TN_CONTRACT_MIP360_INF) n = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
        else n = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float far_branch = (2.0f - 1.0f / n) * p[i] / n;
            c[i] = ((n <= 1.0f) ? p[i] : far_branch) / 2.0f;
        }
    }
    return inside;
}

// One candidate: returns keep flag; fills contracted coords and the step size.
template <int MARCH, int CONTRACT>
__device__ __forceinline__ bool candidate(const SamplerArgs &a, const float o[3], const float d[3], float t_min,
                                          int64_t ray, int k, float c[3], float &delta)
{
    float t;
    march_t<MARCH>(a, t_min, k, t, delta);
    if (a.jitter) {
        t = t + a.jitter[ray * a.n_samples + k] * delta;    // core.py:173
    } else if (a.use_rng) {
        t = t + tn::uniform01(a.seed, (uint64_t)ray * (uint64_t)a.n_samples + (uint64_t)k) * delta;
    }
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = o[i] + d[i] * t;     // core.py:174
    if (!contract<CONTRACT>(a, p, c)) return false;
    if (a.coarse != nullptr && surely_empty(a, c[0], c[1], c[2])) return false;
    return trilinear(a.grid, a.gd, a.gh, a.gw, c[0], c[1], c[2]) > a.threshold;   // core.py:156
}

template <int MARCH>
__global__ void march_rays_kernel(SamplerArgs a, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                  int64_t n_rays, float *__restrict__ t_values, float *__restrict__ step_sizes)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * a.n_samples) return;
    const int64_t ray = i / a.n_samples;
    const int k = (int)(i % a.n_samples);
    float t_min = 0.0f;
    if constexpr (MARCH == TN_MARCH_AABB) {
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = rays_o[3 * ray + c]; d[c] = rays_d[3 * ray + c]; }
        t_min = aabb_t_min(a, o, d);
    }
    float t, delta;
    march_t<MARCH>(a, t_min, k, t, delta);
    t_values[i] = t;
    step_sizes[i] = delta;
}

template <int CONTRACT>
__global__ void contract_kernel(SamplerArgs a, const float *__restrict__ coords, int64_t n, float *__restrict__ out,
                                uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
    float c[3];
    const bool inside = contract<CONTRACT>(a, p, c);
    out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
    if (mask) mask[i] = inside ? 1 : 0;
}

template <int MARCH, int CONTRACT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void sample_mask_kernel(
    SamplerArgs a, const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t n_rays,
    uint64_t *__restrict__ maskbits, int32_t *__restrict__ counts)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = rays_o[3 * ray + i]; d[i] = rays_d[3 * ray + i]; }
    const float t_min = (MARCH == TN_MARCH_AABB) ? aabb_t_min(a, o, d) : 0.0f;
    const int n_chunks = (a.n_samples + 63) >> 6;
    // Box marcher + box contraction: behind the ray's exit every candidate fails the in-box test (core.py:29), so the chunks there
    // need no evaluation.  Exact, not approximate: t_k = fl(t_min + fl(k step)) (+ a jitter >= 0) and p_i = fl(o_i + fl(d_i t)) are
    // monotone in k -- rounding is monotone -- so once the UNJITTERED first candidate of a chunk lies beyond a face along an axis the
    // ray moves outwards on, every later candidate does.  The chunk is guessed from the slab exit and verified with the kernel's
    // own arithmetic; a failed check just evaluates everything.  (Explicit jitter tables may hold anything: not used with them.)
    int n_active = n_chunks;
    if constexpr (MARCH == TN_MARCH_AABB && CONTRACT == TN_CONTRACT_AABB) {
        if (a.jitter == nullptr) {
            float t_exit = 3.0e38f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float den = (d[i] == 0.0f) ? (d[i] + 1e-9f) : d[i];
                t_exit = fminf(t_exit, fmaxf((a.lo[i] - o[i]) / den, (a.hi[i] - o[i]) / den));
            }
            const float kf = (t_exit - t_min) / a.step + 2.0f;                  // two steps of slack: the check below decides
            if (kf >= 0.0f && kf < (float)a.n_samples) {
                const int ch0 = ((int)kf + 63) >> 6;
                if (ch0 < n_chunks) {
                    float t0, delta0;
                    march_t<MARCH>(a, t_min, ch0 * 64, t0, delta0);
                    bool beyond = false;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float pi = o[i] + d[i] * t0;                       // (the candidate's own expression, core.py:174)
                        beyond = beyond || (d[i] > 0.0f && pi > a.hi[i]) || (d[i] < 0.0f && pi < a.lo[i]);
                    }
                    if (beyond) n_active = ch0;
                }
            }
        }
    }
    int count = 0;
    for (int ch = 0; ch < n_active; ++ch) {
        const int k = ch * 64 + lane;
        float c[3], delta;
        bool keep = false;
        if (k < a.n_samples) keep = candidate<MARCH, CONTRACT>(a, o, d, t_min, ray, k, c, delta);
        const uint64_t m = __ballot(keep);
        if (lane == 0) maskbits[ray * n_chunks + ch] = m;
        count += __popcll(m);
    }
    if (n_active + lane < n_chunks) maskbits[ray * n_chunks + n_active + lane] = 0;      // (n_chunks <= 64 for S <= 4096)
    for (int ch = n_active + 64 + lane; ch < n_chunks; ch += 64) maskbits[ray * n_chunks + ch] = 0;
    if (lane == 0) counts[ray] = count;
}

// Exclusive int32 scan of the per-ray counts (reference core.py:179-181).  One workgroup of 1024
// threads sweeps the rays in tiles of 4096 (4 consecutive rays per thread) with a running carry: R is
// 10^3..10^6, the pass moves 12 B/ray; it sits right behind the step's host read-back, so its latency counts.
__global__ __launch_bounds__(1024) void sample_scan_kernel(
    const int32_t *__restrict__ counts, int64_t n_rays, const int32_t *__restrict__ base_offset,
    int32_t *__restrict__ info, int32_t *__restrict__ total)
{
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry_s;
    const int lane = tn::lane_id(), wave = threadIdx.x >> 6;
    const int32_t base = base_offset ? base_offset[0] : 0;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t tile = 0; tile < n_rays; tile += 4096) {
        const int64_t r0 = tile + 4 * (int64_t)threadIdx.x;
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (r0 + u < n_rays) ? counts[r0 + u] : 0;
        const int32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
        int32_t v = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        if (lane == 63) wave_tot[wave] = v;
        __syncthreads();
        int32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
        const int32_t carry = carry_s;
        int32_t run = base + carry + wave_off + v - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 + u < n_rays) { info[2 * (r0 + u)] = run; info[2 * (r0 + u) + 1] = c[u]; }
            run += c[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + v;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) total[0] = carry_s;
}

// Multi-workgroup form of the scan: workgroup b owns rays [4096 b, 4096 (b + 1)) and reads the counts below its range itself
// (b 16-byte loads per thread out of L2) instead of waiting for a carry -- no inter-workgroup traffic, no flags, one launch;
// integer sums, so the result does not depend on the order.  counts must be 16-byte aligned (the launcher checks).
__global__ __launch_bounds__(1024) void sample_scan_mb_kernel(
    const int32_t *__restrict__ counts, int64_t n_rays, const int32_t *__restrict__ base_offset,
    int32_t *__restrict__ info, int32_t *__restrict__ total)
{
    __shared__ int32_t wave_tot[16], pre_tot[16];
    const int lane = tn::lane_id(), wave = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.x * 4096;
    int32_t pre = 0;
    const int4 *c4 = reinterpret_cast<const int4 *>(counts);
    for (int64_t q = threadIdx.x; q < (lo >> 2); q += 1024) { const int4 v = c4[q]; pre += (v.x + v.y) + (v.z + v.w); }
    const int64_t r0 = lo + 4 * (int64_t)threadIdx.x;
    int32_t c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (r0 + u < n_rays) ? counts[r0 + u] : 0;
    const int32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
    int32_t v = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 63) wave_tot[wave] = v;
    if (lane == 0) pre_tot[wave] = pre;
    __syncthreads();
    int32_t below = 0, wave_off = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { below += pre_tot[w]; wave_off += w < wave ? wave_tot[w] : 0; }
    int32_t run = (base_offset ? base_offset[0] : 0) + below + wave_off + v - mine;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (r0 + u < n_rays) { info[2 * (r0 + u)] = run; info[2 * (r0 + u) + 1] = c[u]; }
        run += c[u];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) total[0] = below + wave_off + v;
}

// Dynamic-batch rule AND the scan of every candidate ray in one launch (tn_batch_plan_scan): workgroup b owns loader batch b.
// Its waves first reduce the batches below b (wave w: batches w, w + 16, ...; the counts are L2-resident, written by the mask
// pass just before), so every workgroup knows its own start without waiting for another one; then it scans its own rays.  The
// last workgroup holds every batch sum and evaluates the sequential rule on one lane.  <= MAX_PLAN_BATCHES workgroups: the
// redundant reads grow with the square of the batch count (48 batches of 1024 rays: 4.7 MB out of L2 in total).
constexpr int MAX_PLAN_BATCHES = 256;
__global__ __launch_bounds__(1024) void batch_plan_scan_kernel(const int32_t *__restrict__ counts, int64_t n_rays, int32_t batch_size,
                                                               int64_t target, int32_t *__restrict__ plan, int32_t *__restrict__ info)
{
    __shared__ int32_t sums[MAX_PLAN_BATCHES];
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry_s;
    const int lane = tn::lane_id(), wave = threadIdx.x >> 6;
    const int b = blockIdx.x, n_batches = gridDim.x;
    for (int j = wave; j < b; j += 16) {
        int32_t s = 0;
        const int64_t lo = (int64_t)j * batch_size, hi = lo + batch_size;             // j < b: a full batch
        for (int64_t r = lo + lane; r < hi; r += 64) s += counts[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sums[j] = s;
    }
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    int32_t below = 0;
    for (int j = 0; j < b; ++j) below += sums[j];                                     // LDS broadcast reads
    const int64_t lo = (int64_t)b * batch_size;
    const int64_t hi = lo + batch_size < n_rays ? lo + batch_size : n_rays;
    for (int64_t tile = lo; tile < hi; tile += 4096) {
        const int64_t r0 = tile + 4 * (int64_t)threadIdx.x;
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (r0 + u < hi) ? counts[r0 + u] : 0;
        const int32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
        int32_t v = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        if (lane == 63) wave_tot[wave] = v;
        __syncthreads();
        int32_t wave_off = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) wave_off += w < wave ? wave_tot[w] : 0;
        const int32_t carry = carry_s;
        int32_t run = below + carry + wave_off + v - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 + u < hi) { info[2 * (r0 + u)] = run; info[2 * (r0 + u) + 1] = c[u]; }
            run += c[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + v;
        __syncthreads();
    }
    if (b == n_batches - 1 && threadIdx.x == 0) {
        sums[b] = carry_s;
        int64_t cur = 0;
        int k = 0, tripped = 0;
        while (k < n_batches) {
            cur += sums[k];
            ++k;
            const int64_t projected = (int64_t)((double)cur * (1.0 + 1.0 / (double)k));     // run.py:240
            if (projected >= target) { tripped = 1; break; }
        }
        const int64_t R = (int64_t)k * batch_size < n_rays ? (int64_t)k * batch_size : n_rays;
        plan[0] = k; plan[1] = (int32_t)cur; plan[2] = (int32_t)R; plan[3] = tripped;
    }
}

// reference run.py:215-244 -- the dynamic-batch projection rule evaluated on the device: per-loader-batch sums
// by wave reductions, then the (inherently sequential, <= 4096 steps) rule on one lane.
__global__ __launch_bounds__(1024) void batch_plan_kernel(const int32_t *__restrict__ counts, int64_t n_rays,
                                                          int32_t batch_size, int64_t target, int32_t *__restrict__ plan)
{
    __shared__ int32_t sums[4096];
    const int lane = tn::lane_id(), wave = threadIdx.x >> 6;
    const int n_batches = (int)((n_rays + batch_size - 1) / batch_size);
    for (int b = wave; b < n_batches; b += 16) {
        int32_t s = 0;
        const int64_t lo = (int64_t)b * batch_size;
        const int64_t hi = lo + batch_size < n_rays ? lo + batch_size : n_rays;
        for (int64_t r = lo + lane; r < hi; r += 64) s += counts[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sums[b] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t cur = 0;
        int k = 0, tripped = 0;
        while (k < n_batches) {
            cur += sums[k];
            ++k;
            const int64_t projected = (int64_t)((double)cur * (1.0 + 1.0 / (double)k));     // run.py:240
            if (projected >= target) { tripped = 1; break; }
        }
        const int64_t R = (int64_t)k * batch_size < n_rays ? (int64_t)k * batch_size : n_rays;
        plan[0] = k; plan[1] = (int32_t)cur; plan[2] = (int32_t)R; plan[3] = tripped;
    }
}

template <int MARCH, int CONTRACT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void sample_pack_kernel(
    SamplerArgs a, const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t n_rays,
    const uint64_t *__restrict__ maskbits, const int32_t *__restrict__ info, const int32_t *__restrict__ base_offset,
    float *__restrict__ packed, int32_t *__restrict__ ray_ids, float *__restrict__ steps, int64_t capacity)
{
    const int lane = tn::lane_id();
    const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int count = info[2 * ray + 1];
    if (count == 0) return;
    int64_t out = (int64_t)info[2 * ray] - (base_offset ? base_offset[0] : 0);
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = rays_o[3 * ray + i]; d[i] = rays_d[3 * ray + i]; }
    const float t_min = (MARCH == TN_MARCH_AABB) ? aabb_t_min(a, o, d) : 0.0f;
    const int n_chunks = (a.n_samples + 63) >> 6;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const uint64_t m = maskbits[ray * n_chunks + ch];
        if (m == 0) continue;
        if ((m >> lane) & 1ull) {
            const int k = ch * 64 + lane;
            float c[3], delta;
            (void)candidate<MARCH, CONTRACT>(a, o, d, t_min, ray, k, c, delta);
            const int64_t row = out + tn::rank_below(m);
            if (row < capacity) {
                float *p = packed + 7 * row;
                p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
                p[3] = d[0]; p[4] = d[1]; p[5] = d[2];
                p[6] = delta;
                if (ray_ids) ray_ids[row] = (int32_t)ray;
                if (steps) steps[row] = delta;
            }
        }
        out += __popcll(m);
    }
}

__global__ void occupancy_query_kernel(const float *__restrict__ grid, int D, int H, int W,
                                       const float *__restrict__ coords, int64_t n, float threshold,
                                       uint8_t *__restrict__ out, float *__restrict__ values)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = trilinear(grid, D, H, W, coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
    if (out) out[i] = v > threshold ? 1 : 0;
    if (values) values[i] = v;
}

// reference core.py:109-119,136
__global__ void occupancy_slice_coords_kernel(int D, int H, int W, int slice, const float *__restrict__ jitter,
                                              uint64_t seed, float *__restrict__ coords)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int h = (int)(i / W), w = (int)(i % W);
    const float idx[3] = {(float)w, (float)h, (float)slice};     // flipped to (x,y,z)
    const float size[3] = {(float)D, (float)H, (float)W};        // NOT flipped in the reference
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float u = jitter ? jitter[3 * i + c]
                               : tn::uniform01(seed, ((uint64_t)slice * (uint64_t)H * W + (uint64_t)i) * 3 + c);
        coords[3 * i + c] = -1.0f + 2.0f * (idx[c] + u) / size[c];
    }
}

// reference core.py:137-143
__global__ void occupancy_apply_kernel(float *__restrict__ cells, const float *__restrict__ sigmas, int64_t n,
                                       float step, float threshold, float decay)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float alpha = 1.0f - expf(-sigmas[i] * step);
    cells[i] = alpha > threshold ? 1.0f : decay * cells[i];
}

__global__ __launch_bounds__(256) void occupancy_stats_kernel(const float *__restrict__ grid, int64_t n, float threshold,
                                                              double *__restrict__ stats)
{
    double s = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = grid[i];
        s += (double)v;
        c += v > threshold ? 1.0 : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    __shared__ double ss[4], cc[4];
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; cc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[0], ss[0] + ss[1] + ss[2] + ss[3]);
        atomicAdd(&stats[1], cc[0] + cc[1] + cc[2] + cc[3]);
    }
}

__global__ void occupancy_coarsen_kernel(const float *__restrict__ grid, int D, int H, int W, float *__restrict__ coarse)
{
    const int cw = (W + 3) >> 2, ch = (H + 3) >> 2, cd = (D + 3) >> 2;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (int64_t)cw * ch * cd) return;
    const int bx = (int)(b % cw), by = (int)((b / cw) % ch), bz = (int)(b / ((int64_t)cw * ch));
    float m = 0.0f;                                        // out-of-range taps contribute 0 (zeros padding)
    bool first = true;
    for (int z = 4 * bz; z <= min(4 * bz + 4, D - 1); ++z)
        for (int y = 4 * by; y <= min(4 * by + 4, H - 1); ++y)
            for (int x = 4 * bx; x <= min(4 * bx + 4, W - 1); ++x) {
                const float v = grid[((int64_t)z * H + y) * W + x];
                m = first ? v : fmaxf(m, v);               // (fmaxf drops NaNs; a NaN grid is broken either way)
                first = false;
            }
    coarse[b] = fmaxf(m, 0.0f);
}

int make_args(const tn_sampler_desc *d, SamplerArgs &a, bool need_grid = true)
{
    TN_REQUIRE(d, TN_E_NULL, "sampler: null descriptor");
    TN_REQUIRE(d->n_samples > 0, TN_E_SIZE, "sampler: n_samples must be positive");
    TN_REQUIRE(!need_grid || (d->grid && d->grid_d > 0 && d->grid_h > 0 && d->grid_w > 0), TN_E_NULL, "sampler: occupancy grid missing");
    TN_REQUIRE(d->marcher == TN_MARCH_AABB || d->marcher == TN_MARCH_UNBOUNDED, TN_E_CONFIG, "sampler: unknown marcher");
    TN_REQUIRE(d->contraction >= TN_CONTRACT_AABB && d->contraction <= TN_CONTRACT_MIP360_L2, TN_E_CONFIG, "sampler: unknown contraction");
    TN_REQUIRE(d->marcher != TN_MARCH_UNBOUNDED || (d->t_table && d->delta_table), TN_E_NULL, "sampler: unbounded marcher needs t/delta tables");
    a.marcher = d->marcher; a.contraction = d->contraction; a.n_samples = d->n_samples;
    a.gd = d->grid_d; a.gh = d->grid_h; a.gw = d->grid_w;
    for (int i = 0; i < 3; ++i) { a.lo[i] = d->aabb[i]; a.hi[i] = d->aabb[3 + i]; }
    a.near = d->near; a.far = d->far; a.step = d->step_size; a.threshold = d->threshold;
    a.t_table = d->t_table; a.delta_table = d->delta_table; a.grid = d->grid; a.jitter = d->jitter;
    a.seed = d->seed; a.use_rng = d->use_rng; a.coarse = d->coarse;
    a.pow2 = 1;
    for (int i = 0; i < 3; ++i) {
        const volatile float ext = a.hi[i] - a.lo[i];            // the device's fp32 subtraction
        int e = 0;
        const float m = frexpf(ext, &e);
        a.inv_ext[i] = 1.0f / ext;
        if (!(m == 0.5f && e > -100 && e < 100)) a.pow2 = 0;     // (also rejects 0, negatives, inf, NaN)
    }
    return TN_OK;
}

inline unsigned ray_blocks(int64_t n_rays) { return (unsigned)((n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

// compile-time specialisation of the (marcher, contraction) pair: no per-candidate branching
#define TN_DISPATCH_MC(a, ...)                                                                        \
    do {                                                                                              \
        const int key_ = (a).marcher * 3 + (a).contraction;                                           \
        switch (key_) {                                                                               \
        case 0: { constexpr int M = 0, Cn = 0; __VA_ARGS__; } break;                                         \
        case 1: { constexpr int M = 0, Cn = 1; __VA_ARGS__; } break;                                         \
        case 2: { constexpr int M = 0, Cn = 2; __VA_ARGS__; } break;                                         \
        case 3: { constexpr int M = 1, Cn = 0; __VA_ARGS__; } break;                                         \
        case 4: { constexpr int M = 1, Cn = 1; __VA_ARGS__; } break;                                         \
        default: { constexpr int M = 1, Cn = 2; __VA_ARGS__; } break;                                        \
        }                                                                                             \
    } while (0)

}  // namespace

extern "C" int tn_march_rays(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d, int64_t n_rays,
                             float *t_values, float *step_sizes, void *stream)
{
    SamplerArgs a;
    if (int rc = make_args(desc, a, false)) return rc;
    TN_REQUIRE(n_rays >= 0, TN_E_SIZE, "tn_march_rays: negative n_rays");
    if (n_rays == 0) return TN_OK;
    TN_REQUIRE(rays_o && rays_d && t_values && step_sizes, TN_E_NULL, "tn_march_rays: null pointer");
    const int64_t n = n_rays * a.n_samples;
    if (a.marcher == TN_MARCH_AABB)
        hipLaunchKernelGGL(march_rays_kernel<TN_MARCH_AABB>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           a, rays_o, rays_d, n_rays, t_values, step_sizes);
    else
        hipLaunchKernelGGL(march_rays_kernel<TN_MARCH_UNBOUNDED>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           a, rays_o, rays_d, n_rays, t_values, step_sizes);
    return tn::check_launch("march_rays_kernel");
}

extern "C" int tn_contract(const tn_sampler_desc *desc, const float *coords, int64_t n, float *coords_out, uint8_t *mask,
                           void *stream)
{
    SamplerArgs a;
    if (int rc = make_args(desc, a, false)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_contract: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(coords && coords_out, TN_E_NULL, "tn_contract: null pointer");
    const dim3 grid((unsigned)((n + 255) / 256));
    if (a.contraction == TN_CONTRACT_AABB)
        hipLaunchKernelGGL(contract_kernel<TN_CONTRACT_AABB>, grid, dim3(256), 0, (hipStream_t)stream, a, coords, n, coords_out, mask);
    else if (a.contraction == TN_CONTRACT_MIP360_INF)
        hipLaunchKernelGGL(contract_kernel<TN_CONTRACT_MIP360_INF>, grid, dim3(256), 0, (hipStream_t)stream, a, coords, n, coords_out, mask);
    else
        hipLaunchKernelGGL(contract_kernel<TN_CONTRACT_MIP360_L2>, grid, dim3(256), 0, (hipStream_t)stream, a, coords, n, coords_out, mask);
    return tn::check_launch("contract_kernel");
}

extern "C" int tn_sample_mask(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d, int64_t n_rays,
                              uint64_t *maskbits, int32_t *counts, void *stream)
{
    SamplerArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n_rays >= 0, TN_E_SIZE, "tn_sample_mask: negative n_rays");
    if (n_rays == 0) return TN_OK;
    TN_REQUIRE(rays_o && rays_d && maskbits && counts, TN_E_NULL, "tn_sample_mask: null pointer");
    TN_DISPATCH_MC(a, sample_mask_kernel<M, Cn><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(
                          a, rays_o, rays_d, n_rays, maskbits, counts));
    return tn::check_launch("sample_mask_kernel");
}

extern "C" int tn_sample_scan(const int32_t *counts, int64_t n_rays, const int32_t *base_offset, int32_t *info,
                              int32_t *total, void *stream)
{
    TN_REQUIRE(n_rays >= 0, TN_E_SIZE, "tn_sample_scan: negative n_rays");
    TN_REQUIRE(n_rays == 0 || (counts && info), TN_E_NULL, "tn_sample_scan: null pointer");
    if (n_rays > 4096 && n_rays <= ((int64_t)1 << 21) && (reinterpret_cast<uintptr_t>(counts) & 15) == 0) {
        hipLaunchKernelGGL(sample_scan_mb_kernel, dim3((unsigned)((n_rays + 4095) / 4096)), dim3(1024), 0, (hipStream_t)stream, counts, n_rays,
                           base_offset, info, total);
        return tn::check_launch("sample_scan_mb_kernel");
    }
    hipLaunchKernelGGL(sample_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n_rays, base_offset, info, total);
    return tn::check_launch("sample_scan_kernel");
}

extern "C" int tn_batch_plan(const int32_t *counts, int64_t n_rays, int32_t batch_size, int64_t target, int32_t *plan,
                             void *stream)
{
    TN_REQUIRE(n_rays >= 0 && batch_size > 0 && target >= 0, TN_E_SIZE, "tn_batch_plan: bad size");
    TN_REQUIRE((n_rays + batch_size - 1) / batch_size <= 4096, TN_E_SIZE, "tn_batch_plan: more than 4096 loader batches");
    TN_REQUIRE(plan && (n_rays == 0 || counts), TN_E_NULL, "tn_batch_plan: null pointer");
    batch_plan_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(counts, n_rays, batch_size, target, plan);
    return tn::check_launch("batch_plan_kernel");
}

extern "C" int tn_batch_plan_scan(const int32_t *counts, int64_t n_rays, int32_t batch_size, int64_t target, int32_t *plan,
                                  int32_t *info, void *stream)
{
    TN_REQUIRE(n_rays >= 0 && batch_size > 0 && target >= 0, TN_E_SIZE, "tn_batch_plan_scan: bad size");
    const int64_t n_batches = (n_rays + batch_size - 1) / batch_size;
    TN_REQUIRE(n_batches <= 4096, TN_E_SIZE, "tn_batch_plan_scan: more than 4096 loader batches");
    TN_REQUIRE(plan && (n_rays == 0 || (counts && info)), TN_E_NULL, "tn_batch_plan_scan: null pointer");
    if (n_batches >= 1 && n_batches <= MAX_PLAN_BATCHES) {
        batch_plan_scan_kernel<<<dim3((unsigned)n_batches), dim3(1024), 0, (hipStream_t)stream>>>(counts, n_rays, batch_size, target, plan, info);
        return tn::check_launch("batch_plan_scan_kernel");
    }
    if (int rc = tn_batch_plan(counts, n_rays, batch_size, target, plan, stream)) return rc;
    return n_rays ? tn_sample_scan(counts, n_rays, nullptr, info, nullptr, stream) : TN_OK;
}

extern "C" int tn_sample_pack(const tn_sampler_desc *desc, const float *rays_o, const float *rays_d, int64_t n_rays,
                              const uint64_t *maskbits, const int32_t *info, const int32_t *base_offset, float *packed,
                              int32_t *ray_ids, float *steps, int64_t capacity, void *stream)
{
    SamplerArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n_rays >= 0 && capacity >= 0, TN_E_SIZE, "tn_sample_pack: negative size");
    if (n_rays == 0 || capacity == 0) return TN_OK;
    TN_REQUIRE(rays_o && rays_d && maskbits && info && packed, TN_E_NULL, "tn_sample_pack: null pointer");
    TN_DISPATCH_MC(a, sample_pack_kernel<M, Cn><<<dim3(ray_blocks(n_rays)), dim3(WAVES_PER_BLOCK * 64), 0, (hipStream_t)stream>>>(
                          a, rays_o, rays_d, n_rays, maskbits, info, base_offset, packed, ray_ids, steps, capacity));
    return tn::check_launch("sample_pack_kernel");
}

extern "C" int tn_occupancy_query(const float *grid, int D, int H, int W, const float *coords, int64_t n, float threshold,
                                  uint8_t *out, float *values, void *stream)
{
    TN_REQUIRE(n >= 0 && D > 0 && H > 0 && W > 0, TN_E_SIZE, "tn_occupancy_query: bad size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(grid && coords && (out || values), TN_E_NULL, "tn_occupancy_query: null pointer");
    hipLaunchKernelGGL(occupancy_query_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grid, D, H, W, coords, n, threshold, out, values);
    return tn::check_launch("occupancy_query_kernel");
}

extern "C" int tn_occupancy_slice_coords(int D, int H, int W, int slice, const float *jitter, uint64_t seed, float *coords,
                                         void *stream)
{
    TN_REQUIRE(D > 0 && H > 0 && W > 0 && slice >= 0 && slice < D, TN_E_SIZE, "tn_occupancy_slice_coords: bad size");
    TN_REQUIRE(coords, TN_E_NULL, "tn_occupancy_slice_coords: null pointer");
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(occupancy_slice_coords_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       D, H, W, slice, jitter, seed, coords);
    return tn::check_launch("occupancy_slice_coords_kernel");
}

extern "C" int tn_occupancy_apply(float *grid_cells, const float *sigmas, int64_t n, float step_size, float threshold,
                                  float decay, void *stream)
{
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_occupancy_apply: negative size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(grid_cells && sigmas, TN_E_NULL, "tn_occupancy_apply: null pointer");
    hipLaunchKernelGGL(occupancy_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grid_cells, sigmas, n, step_size, threshold, decay);
    return tn::check_launch("occupancy_apply_kernel");
}

extern "C" int tn_occupancy_stats(const float *grid, int64_t n, float threshold, double *stats, void *stream)
{
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_occupancy_stats: negative size");
    TN_REQUIRE(grid && stats, TN_E_NULL, "tn_occupancy_stats: null pointer");
    hipError_t e = hipMemsetAsync(stats, 0, 2 * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) { tn::set_error("tn_occupancy_stats: memset: %s", hipGetErrorString(e)); return (int)e; }
    if (n == 0) return TN_OK;
    const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(occupancy_stats_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, (hipStream_t)stream,
                       grid, n, threshold, stats);
    return tn::check_launch("occupancy_stats_kernel");
}

extern "C" int tn_occupancy_coarsen(const float *grid, int D, int H, int W, float *coarse, void *stream)
{
    TN_REQUIRE(D > 0 && H > 0 && W > 0, TN_E_SIZE, "tn_occupancy_coarsen: bad size");
    TN_REQUIRE(grid && coarse, TN_E_NULL, "tn_occupancy_coarsen: null pointer");
    const int64_t nb = (int64_t)((D + 3) / 4) * ((H + 3) / 4) * ((W + 3) / 4);
    occupancy_coarsen_kernel<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(grid, D, H, W, coarse);
    return tn::check_launch("occupancy_coarsen_kernel");
}
