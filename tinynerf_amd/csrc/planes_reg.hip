// K-Planes total-variation / L1 regularisers (reference src/models.py:115-121,165-181) for
// channel-last planes [H][W][C].  The reference evaluates, per plane, two mse_loss calls on shifted
// strided views (plus their autograd backward): ~0.5 GB of traffic per step over the 126 MiB of
// planes.  Here the forward is one streaming read per plane with an fp64 block
// reduction, and the backward one pass that adds the 5-point stencil of the gradient straight into
// the plane's gradient buffer.  HBM-bound: fwd 4 B/element, bwd 12 B/element.
#include "tn_common.h"
#include <algorithm>
#include <math.h>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void plane_reg_fwd_kernel(const float *__restrict__ p, int H, int W, int C4,
                                                            double *__restrict__ sums)
{
    const int64_t total = (int64_t)H * W * C4;
    double sy = 0.0, sx = 0.0, sl = 0.0;
    const f4 *q = reinterpret_cast<const f4 *>(p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t texel = i / C4;
        const int x = (int)(texel % W), y = (int)(texel / W);
        const f4 v = q[i];
        sl += (double)(fabsf(v[0]) + fabsf(v[1]) + fabsf(v[2]) + fabsf(v[3]));
        if (y + 1 < H) { const f4 d = q[i + (int64_t)W * C4] - v; sy += (double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]); }
        if (x + 1 < W) { const f4 d = q[i + C4] - v; sx += (double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sy += __shfl_xor(sy, o, 64); sx += __shfl_xor(sx, o, 64); sl += __shfl_xor(sl, o, 64); }
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sy; red[1][threadIdx.x >> 6] = sx; red[2][threadIdx.x >> 6] = sl; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void plane_reg_bwd_kernel(const float *__restrict__ p, int H, int W, int C4, float cy, float cx,
                                                            float cl1, const float *__restrict__ upstream, float *__restrict__ grad)
{
    const int64_t total = (int64_t)H * W * C4;
    const float up = upstream[0];
    const f4 *q = reinterpret_cast<const f4 *>(p);
    f4 *g = reinterpret_cast<f4 *>(grad);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t texel = i / C4;
        const int x = (int)(texel % W), y = (int)(texel / W);
        const f4 v = q[i];
        f4 dy = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
        if (y > 0) dy += v - q[i - (int64_t)W * C4];
        if (y + 1 < H) dy -= q[i + (int64_t)W * C4] - v;
        if (x > 0) dx += v - q[i - C4];
        if (x + 1 < W) dx -= q[i + C4] - v;
        f4 r = (dy * (2.0f * cy) + dx * (2.0f * cx));
        if (cl1 != 0.0f) {
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] += cl1 * (v[c] > 0.f ? 1.f : (v[c] < 0.f ? -1.f : 0.f));
        }
        g[i] += r * up;
    }
}

// torch.optim.Adam (no amsgrad, coupled weight decay) for one tensor, element-wise, any memory layout
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, int zero_grad)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        if (4 * i + 3 < n) {
            f4 pv = reinterpret_cast<f4 *>(p)[i], gv = reinterpret_cast<f4 *>(g)[i];
            f4 mv = reinterpret_cast<f4 *>(m)[i], vv = reinterpret_cast<f4 *>(v)[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gg = gv[c] + wd * pv[c];
                mv[c] = mv[c] + (gg - mv[c]) * (1.0f - b1);            // lerp form, as torch
                vv[c] = b2 * vv[c] + (1.0f - b2) * gg * gg;
                const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
                pv[c] = pv[c] - (lr / bc1) * (mv[c] / denom);
            }
            reinterpret_cast<f4 *>(p)[i] = pv; reinterpret_cast<f4 *>(m)[i] = mv; reinterpret_cast<f4 *>(v)[i] = vv;
            if (zero_grad) reinterpret_cast<f4 *>(g)[i] = f4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int64_t e = 4 * i; e < n; ++e) {
                const float gg = g[e] + wd * p[e];
                m[e] = m[e] + (gg - m[e]) * (1.0f - b1);
                v[e] = b2 * v[e] + (1.0f - b2) * gg * gg;
                p[e] = p[e] - (lr / bc1) * (m[e] / (sqrtf(v[e]) / bc2_sqrt + eps));
                if (zero_grad) g[e] = 0.0f;
            }
        }
    }
}

// ---- multi-tensor forms: the harness has 9 planes and ~20 parameter tensors; one launch each instead of one per
// tensor removes ~40 launches (and the host time between them) from every step.  blockIdx.y = item.
struct RegItems { tn_plane_reg_item it[TN_MULTI_MAX]; };
struct AdamItems { tn_adam_item it[TN_MULTI_MAX]; };

// fused forward + backward of the regulariser for a constant upstream gradient: the plane is read once, the three
// sums go to fp64 accumulators and the 5-point stencil of the gradient is added to the gradient buffer.
__global__ __launch_bounds__(256) void plane_reg_multi_kernel(RegItems items, float up, double *__restrict__ sums)
{
    const tn_plane_reg_item &t = items.it[blockIdx.y];
    const int W = t.W, H = t.H, C4 = t.C >> 2;
    const int64_t total = (int64_t)H * W * C4;
    const f4 *q = reinterpret_cast<const f4 *>(t.plane);
    f4 *g = reinterpret_cast<f4 *>(t.grad);
    const float cy2 = 2.0f * t.cy, cx2 = 2.0f * t.cx;
    float sy = 0.f, sx = 0.f, sl = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t texel = i / C4;
        const int x = (int)(texel % W), y = (int)(texel / W);
        const f4 v = q[i];
        f4 dy = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
        if (y > 0) dy += v - q[i - (int64_t)W * C4];
        if (x > 0) dx += v - q[i - C4];
        if (y + 1 < H) { const f4 d = q[i + (int64_t)W * C4] - v; dy -= d; sy += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
        if (x + 1 < W) { const f4 d = q[i + C4] - v; dx -= d; sx += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
        sl += fabsf(v[0]) + fabsf(v[1]) + fabsf(v[2]) + fabsf(v[3]);
        if (g != nullptr) {
            f4 r = dy * cy2 + dx * cx2;
            if (t.cl1 != 0.0f) {
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] += t.cl1 * (v[c] > 0.f ? 1.f : (v[c] < 0.f ? -1.f : 0.f));
            }
            g[i] += r * up;
        }
    }
    if (sums == nullptr) return;
    // per-thread partials are fp32 over <= a few hundred texels; everything above that is fp64
    double dsy = sy, dsx = sx, dsl = sl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dsy += __shfl_xor(dsy, o, 64); dsx += __shfl_xor(dsx, o, 64); dsl += __shfl_xor(dsl, o, 64); }
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dsy; red[1][threadIdx.x >> 6] = dsx; red[2][threadIdx.x >> 6] = dsl; }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(&sums[3 * blockIdx.y + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void adam_gate_kernel(int32_t *__restrict__ step, const float *__restrict__ gate)
{
    if (gate[0] > 0.0f) step[0] += 1;
}

// GATED: bias corrections from a device-side step count, nothing but the optional gradient zeroing when the gate is closed
template <bool GATED>
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamItems items, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, int zero_grad, const int32_t *__restrict__ step_dev,
                                                         const float *__restrict__ gate, int32_t *__restrict__ nonfinite)
{
    const tn_adam_item &t = items.it[blockIdx.y];
    bool bad = false;            // an updated parameter that is not finite (see tn_adam_multi_gated: zero_grad bit 1)
    float *__restrict__ p = t.param; float *__restrict__ g = t.grad; float *__restrict__ m = t.exp_avg; float *__restrict__ v = t.exp_avg_sq;
    const int64_t n = t.n, n4 = (n + 3) / 4;
    if constexpr (GATED) {
        if (!(gate[0] > 0.0f)) {
            if (zero_grad)
                for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) g[e] = 0.0f;
            return;
        }
        const double st = (double)step_dev[0];
        bc1 = (float)(1.0 - pow((double)b1, st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        if (4 * i + 3 < n) {
            f4 pv = reinterpret_cast<f4 *>(p)[i], gv = reinterpret_cast<f4 *>(g)[i];
            f4 mv = reinterpret_cast<f4 *>(m)[i], vv = reinterpret_cast<f4 *>(v)[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gg = gv[c] + wd * pv[c];
                mv[c] = mv[c] + (gg - mv[c]) * (1.0f - b1);
                vv[c] = b2 * vv[c] + (1.0f - b2) * gg * gg;
                const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
                pv[c] = pv[c] - (lr / bc1) * (mv[c] / denom);
                bad |= !(fabsf(pv[c]) <= 3.402823466e38f);
            }
            reinterpret_cast<f4 *>(p)[i] = pv; reinterpret_cast<f4 *>(m)[i] = mv; reinterpret_cast<f4 *>(v)[i] = vv;
            if (zero_grad) reinterpret_cast<f4 *>(g)[i] = f4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int64_t e = 4 * i; e < n; ++e) {
                const float gg = g[e] + wd * p[e];
                m[e] = m[e] + (gg - m[e]) * (1.0f - b1);
                v[e] = b2 * v[e] + (1.0f - b2) * gg * gg;
                p[e] = p[e] - (lr / bc1) * (m[e] / (sqrtf(v[e]) / bc2_sqrt + eps));
                bad |= !(fabsf(p[e]) <= 3.402823466e38f);
                if (zero_grad) g[e] = 0.0f;
            }
        }
    }
    if (nonfinite != nullptr && bad) nonfinite[0] = 1;          // (plain stores of one value: no atomic needed)
}

// Adam with the K-Planes regulariser folded in (harness): for plane tensors the total-variation / L1 gradient is built
// from the CURRENT values while the update is written to a second buffer (the caller swaps the two), so the planes, their
// gradients and both moments are streamed exactly once per step instead of once for the regulariser and once for Adam.
struct AdamRegItems { tn_adam_reg_item it[TN_MULTI_MAX / 2]; };

__global__ __launch_bounds__(256) void adam_reg_multi_kernel(AdamRegItems items, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                             float bc2_sqrt, int zero_grad, float up, double *__restrict__ sums)
{
    const tn_adam_reg_item &t = items.it[blockIdx.y];
    const float *__restrict__ p = t.param; float *__restrict__ po = t.param_out;
    float *__restrict__ g = t.grad; float *__restrict__ m = t.exp_avg; float *__restrict__ v = t.exp_avg_sq;
    const int64_t n = t.n, n4 = (n + 3) / 4;
    const bool reg = t.H > 0;
    const int W = t.W, H = t.H, C4 = t.C >> 2;
    const float cy2 = 2.0f * t.cy, cx2 = 2.0f * t.cx;
    float sy = 0.f, sx = 0.f, sl = 0.f;
    // sharded pass: this rank's rows of the plane as a range of float4 indices (everything, unless row1 > 0)
    const int64_t own0 = (reg && t.row1 > 0) ? (int64_t)t.row0 * W * C4 : 0, own1 = (reg && t.row1 > 0) ? (int64_t)t.row1 * W * C4 : n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < own0 || i >= own1) {                            // another rank's rows: only the gradient buffer is cleared
            if (zero_grad) reinterpret_cast<f4 *>(g)[i] = f4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        if (4 * i + 3 < n) {
            const f4 *q = reinterpret_cast<const f4 *>(p);
            f4 pv = q[i], gv = reinterpret_cast<f4 *>(g)[i];
            f4 mv = reinterpret_cast<f4 *>(m)[i], vv = reinterpret_cast<f4 *>(v)[i];
            if (reg) {
                const int64_t texel = i / C4;
                const int x = (int)(texel % W), y = (int)(texel / W);
                f4 dy = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
                if (y > 0) dy += pv - q[i - (int64_t)W * C4];
                if (x > 0) dx += pv - q[i - C4];
                if (y + 1 < H) { const f4 d = q[i + (int64_t)W * C4] - pv; dy -= d; sy += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
                if (x + 1 < W) { const f4 d = q[i + C4] - pv; dx -= d; sx += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
                sl += fabsf(pv[0]) + fabsf(pv[1]) + fabsf(pv[2]) + fabsf(pv[3]);
                f4 r = dy * cy2 + dx * cx2;
                if (t.cl1 != 0.0f) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] += t.cl1 * (pv[c] > 0.f ? 1.f : (pv[c] < 0.f ? -1.f : 0.f));
                }
                gv += r * up;                                   // as plane_reg_multi_kernel: g += r * upstream
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gg = gv[c] + wd * pv[c];
                mv[c] = mv[c] + (gg - mv[c]) * (1.0f - b1);
                vv[c] = b2 * vv[c] + (1.0f - b2) * gg * gg;
                const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
                pv[c] = pv[c] - (lr / bc1) * (mv[c] / denom);
            }
            reinterpret_cast<f4 *>(po)[i] = pv; reinterpret_cast<f4 *>(m)[i] = mv; reinterpret_cast<f4 *>(v)[i] = vv;
            if (zero_grad) reinterpret_cast<f4 *>(g)[i] = f4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int64_t e = 4 * i; e < n; ++e) {               // (tails: tensors without a regulariser only)
                const float gg = g[e] + wd * p[e];
                m[e] = m[e] + (gg - m[e]) * (1.0f - b1);
                v[e] = b2 * v[e] + (1.0f - b2) * gg * gg;
                po[e] = p[e] - (lr / bc1) * (m[e] / (sqrtf(v[e]) / bc2_sqrt + eps));
                if (zero_grad) g[e] = 0.0f;
            }
        }
    }
    if (!reg || sums == nullptr) return;
    double dsy = sy, dsx = sx, dsl = sl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dsy += __shfl_xor(dsy, o, 64); dsx += __shfl_xor(dsx, o, 64); dsl += __shfl_xor(dsl, o, 64); }
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dsy; red[1][threadIdx.x >> 6] = dsx; red[2][threadIdx.x >> 6] = dsl; }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(&sums[3 * t.sum_slot + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 8); }

}  // namespace

extern "C" int tn_plane_reg_fwd(const float *plane, int H, int W, int C, double *sums, void *stream)
{
    TN_REQUIRE(H > 0 && W > 0 && C > 0 && (C & 3) == 0, TN_E_SIZE, "tn_plane_reg_fwd: bad shape (C must be a multiple of 4)");
    TN_REQUIRE(plane && sums, TN_E_NULL, "tn_plane_reg_fwd: null pointer");
    TN_REQUIRE(((uintptr_t)plane & 15) == 0, TN_E_ALIGN, "tn_plane_reg_fwd: plane must be 16-byte aligned");
    const int64_t n = (int64_t)H * W * (C / 4);
    plane_reg_fwd_kernel<<<dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream>>>(plane, H, W, C / 4, sums);
    return tn::check_launch("plane_reg_fwd_kernel");
}

extern "C" int tn_plane_reg_bwd(const float *plane, int H, int W, int C, float cy, float cx, float cl1, const float *upstream,
                                float *grad, void *stream)
{
    TN_REQUIRE(H > 0 && W > 0 && C > 0 && (C & 3) == 0, TN_E_SIZE, "tn_plane_reg_bwd: bad shape (C must be a multiple of 4)");
    TN_REQUIRE(plane && upstream && grad, TN_E_NULL, "tn_plane_reg_bwd: null pointer");
    TN_REQUIRE((((uintptr_t)plane | (uintptr_t)grad) & 15) == 0, TN_E_ALIGN, "tn_plane_reg_bwd: buffers must be 16-byte aligned");
    const int64_t n = (int64_t)H * W * (C / 4);
    plane_reg_bwd_kernel<<<dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream>>>(plane, H, W, C / 4, cy, cx, cl1, upstream, grad);
    return tn::check_launch("plane_reg_bwd_kernel");
}

extern "C" int tn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad, void *stream)
{
    TN_REQUIRE(n >= 0 && step >= 1, TN_E_SIZE, "tn_adam_step: bad size / step");
    if (n == 0) return TN_OK;
    TN_REQUIRE(param && grad && exp_avg && exp_avg_sq, TN_E_NULL, "tn_adam_step: null pointer");
    const bool aligned = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));          // host doubles, like torch's python scalars
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    TN_REQUIRE(aligned, TN_E_ALIGN, "tn_adam_step: buffers must be 16-byte aligned");
    const int64_t n4 = (n + 3) / 4;
    adam_kernel<<<dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n4, n, lr, beta1, beta2, eps,
                                                                             weight_decay, bc1, bc2_sqrt, zero_grad);
    return tn::check_launch("adam_kernel");
}

extern "C" int tn_plane_reg_multi(const tn_plane_reg_item *items, int32_t n_items, float upstream, double *sums, void *stream)
{
    TN_REQUIRE(n_items >= 0, TN_E_SIZE, "tn_plane_reg_multi: negative item count");
    TN_REQUIRE(n_items == 0 || items, TN_E_NULL, "tn_plane_reg_multi: null items");
    for (int base = 0; base < n_items; base += TN_MULTI_MAX) {
        RegItems pack;
        const int cnt = std::min(TN_MULTI_MAX, n_items - base);
        int64_t largest = 0;
        for (int i = 0; i < cnt; ++i) {
            const tn_plane_reg_item &t = items[base + i];
            TN_REQUIRE(t.H > 0 && t.W > 0 && t.C > 0 && (t.C & 3) == 0, TN_E_SIZE, "tn_plane_reg_multi: bad shape (C must be a multiple of 4)");
            TN_REQUIRE(t.plane, TN_E_NULL, "tn_plane_reg_multi: null plane");
            TN_REQUIRE((((uintptr_t)t.plane | (uintptr_t)t.grad) & 15) == 0, TN_E_ALIGN, "tn_plane_reg_multi: buffers must be 16-byte aligned");
            pack.it[i] = t;
            largest = std::max<int64_t>(largest, (int64_t)t.H * t.W * (t.C / 4));
        }
        plane_reg_multi_kernel<<<dim3(std::min<unsigned>(blocks_for(largest), 1024), (unsigned)cnt), dim3(256), 0, (hipStream_t)stream>>>(
            pack, upstream, sums ? sums + 3 * base : nullptr);
        if (int rc = tn::check_launch("plane_reg_multi_kernel")) return rc;
    }
    return TN_OK;
}

extern "C" int tn_adam_multi(const tn_adam_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, int32_t zero_grad, void *stream)
{
    TN_REQUIRE(n_items >= 0 && step >= 1, TN_E_SIZE, "tn_adam_multi: bad item count / step");
    TN_REQUIRE(n_items == 0 || items, TN_E_NULL, "tn_adam_multi: null items");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    for (int base = 0; base < n_items; base += TN_MULTI_MAX) {
        AdamItems pack;
        const int cnt = std::min(TN_MULTI_MAX, n_items - base);
        int64_t largest = 0;
        for (int i = 0; i < cnt; ++i) {
            const tn_adam_item &t = items[base + i];
            TN_REQUIRE(t.n >= 0, TN_E_SIZE, "tn_adam_multi: negative size");
            TN_REQUIRE(t.n == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), TN_E_NULL, "tn_adam_multi: null pointer");
            TN_REQUIRE((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0, TN_E_ALIGN,
                       "tn_adam_multi: buffers must be 16-byte aligned");
            pack.it[i] = t;
            largest = std::max<int64_t>(largest, (t.n + 3) / 4);
        }
        if (largest == 0) continue;
        adam_multi_kernel<false><<<dim3(std::min<unsigned>(blocks_for(largest), 1024), (unsigned)cnt), dim3(256), 0, (hipStream_t)stream>>>(
            pack, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, zero_grad, nullptr, nullptr, nullptr);
        if (int rc = tn::check_launch("adam_multi_kernel")) return rc;
    }
    return TN_OK;
}

extern "C" int tn_adam_multi_gated(const tn_adam_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int32_t *step_dev, const float *gate, int32_t zero_grad, void *stream)
{
    TN_REQUIRE(n_items >= 0, TN_E_SIZE, "tn_adam_multi_gated: bad item count");
    TN_REQUIRE((n_items == 0 || items) && step_dev && gate, TN_E_NULL, "tn_adam_multi_gated: null items / step counter / gate");
    adam_gate_kernel<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>(step_dev, gate);
    if (int rc = tn::check_launch("adam_gate_kernel")) return rc;
    for (int base = 0; base < n_items; base += TN_MULTI_MAX) {
        AdamItems pack;
        const int cnt = std::min(TN_MULTI_MAX, n_items - base);
        int64_t largest = 0;
        for (int i = 0; i < cnt; ++i) {
            const tn_adam_item &t = items[base + i];
            TN_REQUIRE(t.n >= 0, TN_E_SIZE, "tn_adam_multi_gated: negative size");
            TN_REQUIRE(t.n == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), TN_E_NULL, "tn_adam_multi_gated: null pointer");
            TN_REQUIRE((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0, TN_E_ALIGN,
                       "tn_adam_multi_gated: buffers must be 16-byte aligned");
            pack.it[i] = t;
            largest = std::max<int64_t>(largest, (t.n + 3) / 4);
        }
        if (largest == 0) continue;
        adam_multi_kernel<true><<<dim3(std::min<unsigned>(blocks_for(largest), 1024), (unsigned)cnt), dim3(256), 0, (hipStream_t)stream>>>(
            pack, lr, beta1, beta2, eps, weight_decay, 1.0f, 1.0f, zero_grad & 1, step_dev, gate, (zero_grad & 2) ? step_dev + 1 : nullptr);
        if (int rc = tn::check_launch("adam_multi_kernel<gated>")) return rc;
    }
    return TN_OK;
}

extern "C" int tn_adam_reg_multi(const tn_adam_reg_item *items, int32_t n_items, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int32_t step, int32_t zero_grad, float upstream, double *sums, void *stream)
{
    TN_REQUIRE(n_items >= 0 && step >= 1, TN_E_SIZE, "tn_adam_reg_multi: bad item count / step");
    TN_REQUIRE(n_items == 0 || items, TN_E_NULL, "tn_adam_reg_multi: null items");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    constexpr int MAXI = TN_MULTI_MAX / 2;
    for (int base = 0; base < n_items; base += MAXI) {
        AdamRegItems pack;
        const int cnt = std::min(MAXI, n_items - base);
        int64_t largest = 0;
        for (int i = 0; i < cnt; ++i) {
            const tn_adam_reg_item &t = items[base + i];
            TN_REQUIRE(t.n >= 0, TN_E_SIZE, "tn_adam_reg_multi: negative size");
            TN_REQUIRE(t.n == 0 || (t.param && t.param_out && t.grad && t.exp_avg && t.exp_avg_sq), TN_E_NULL, "tn_adam_reg_multi: null pointer");
            TN_REQUIRE((((uintptr_t)t.param | (uintptr_t)t.param_out | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0,
                       TN_E_ALIGN, "tn_adam_reg_multi: buffers must be 16-byte aligned");
            if (t.H > 0) {
                TN_REQUIRE(t.W > 0 && t.C > 0 && (t.C & 3) == 0 && (int64_t)t.H * t.W * t.C == t.n, TN_E_SIZE,
                           "tn_adam_reg_multi: plane shape must match n (C a multiple of 4)");
                TN_REQUIRE(t.param_out != t.param, TN_E_CONFIG, "tn_adam_reg_multi: a regularised plane needs a separate output buffer");
                TN_REQUIRE(sums == nullptr || t.sum_slot >= 0, TN_E_SIZE, "tn_adam_reg_multi: bad sum slot");
            }
            // sharded pass: row1 == 0 means every row; anything else must be a non-empty range inside the plane (a mis-sharded caller
            // would otherwise train with frozen rows and no error)
            TN_REQUIRE(t.row1 == 0 ? t.row0 == 0 : (t.H > 0 && t.row0 >= 0 && t.row0 < t.row1 && t.row1 <= t.H), TN_E_SIZE,
                       "tn_adam_reg_multi: rows [row0, row1) must satisfy 0 <= row0 < row1 <= H (row1 == 0, row0 == 0: every row)");
            pack.it[i] = t;
            largest = std::max<int64_t>(largest, (t.n + 3) / 4);
        }
        if (largest == 0) continue;
        adam_reg_multi_kernel<<<dim3(std::min<unsigned>(blocks_for(largest), 1024), (unsigned)cnt), dim3(256), 0, (hipStream_t)stream>>>(
            pack, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, zero_grad, upstream, sums);
        if (int rc = tn::check_launch("adam_reg_multi_kernel")) return rc;
    }
    return TN_OK;
}
