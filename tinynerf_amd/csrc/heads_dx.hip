// d loss / d x of the two width-64 heads behind a wide stack, from their G_0 rows (round 5).
//
// The reference's decoders read the feature stack's output x [n, F] through one Linear(F, 64) each (src/models.py:70-89 on
// VanillaFeatureMLP(10, 256, 8) / Cobafa's 128-wide stack, run.py:131-150), so d loss / d x = W_0c[:, x]^T G_0c + W_0s^T G_0s: a
// [F x 128] [128 x n] product -- 67 GFLOP per 2^20 samples at F = 256, as much as one hidden layer of the stack.  Inside the heads' data-
// gradient chains (mlp_bwd2.hip) it ran on v_mfma_f32_32x32x2_f32 out of fp32 weights in LDS: 0.43 ms of matrix time at best, 0.8 ms
// measured, in launches that also had to hold the hidden layers' weights.  Here it is a launch of its own in the f16x2 arithmetic of the
// layer kernels (mlp_f2_layers.hip): both heads' G_0 rows (written by their chains, which now stop at G_0) are the B operands --
// [feature][32-sample] rows ARE the operand layout, two-term fp16 splits with one power-of-two scale per sample --, the 128 x F
// first-layer weights sit in LDS as hi / lo planes in A-operand order (4 bytes per weight, scaled by the largest |weight|), three
// 32 x 32 x 16 MFMAs per k block: 6 k cycles of matrix pipe per 32-sample tile instead of 33 k.  The launch is then bound by what it
// reads and writes: 16 KB of G_0 rows in, F x 128 B of gradient rows out per tile (x is a hidden activation of the producer when its
// last layer was merged into the heads, TN_MLP_SKIP_LAST: the rows leave multiplied by relu'(x) through the producer's bit rows).
#include "mlp_layers.h"
#include "mlp_f2_heads.h"
#include <algorithm>

namespace {

using namespace tn::mlp;
using tn::f32x16;

struct DxArgs {
    const float *w_a, *w_b;          // first-layer weights [64][ld_a] (x columns from col0_a) and [64][ld_b] (from col0_b)
    int ld_a, col0_a, ld_b, col0_b;
    const float *g_a, *g_b;          // G_0 rows of tile t at g + t * stride  ([64][32] floats)
    int64_t gs_a, gs_b;
    float *out;                      // d loss / d x rows of tile t at out + t * out_stride ([F][32] floats)
    int64_t out_stride;
    const unsigned *mask;            // relu' bit rows (nullptr: none), tile t at mask + t * mask_stride, block ob at + 64 ob
    int64_t mask_stride;
};

// WAVES per workgroup (one workgroup per CU: the weight planes fill LDS): 16 waves without a register prefetch (four per SIMD hide each
// other's round trips, and a wave may only have 64 vector-memory operations in flight: 200 per tile here) against 8 with the next tile's
// rows requested a tile ahead -- TN_DX_WAVES
#ifndef TN_DX_WAVES
#define TN_DX_WAVES 16
#endif
template <int F>
__global__ __launch_bounds__(TN_DX_WAVES * 64) void heads_dx_f2_kernel(DxArgs a, int64_t n)
{
    constexpr int NOB = F / 32, NW = TN_DX_WAVES;
    constexpr bool PREFETCH = NW <= 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16 *whi = reinterpret_cast<_Float16 *>(lds_raw), *wlo = whi + 128 * F;
    float *scratch = reinterpret_cast<float *>(lds_raw + (size_t)2 * 128 * F * 2);
    const int lane = tn::lane_id(), j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- stage W' = [W_a[:, x]; W_b[:, x]] (128 x F) as fp16 hi / lo planes in A-operand order [ob][k block][h][column][8] ----
    auto wv = [&](int nn, int col) { return nn < 64 ? a.w_a[(int64_t)nn * a.ld_a + a.col0_a + col] : a.w_b[(int64_t)(nn - 64) * a.ld_b + a.col0_b + col]; };
    float m = 0.0f;
    for (int e = threadIdx.x; e < 128 * F; e += NW * 64) m = fmaxf(m, fabsf(wv(e / F, e % F)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) scratch[wave] = m;
    __syncthreads();
    float g = 0.0f;
    for (int w = 0; w < NW; ++w) g = fmaxf(g, scratch[w]);
    float s_w, inv_w;
    f2_scales(g, s_w, inv_w);
    for (int e = threadIdx.x; e < 128 * F; e += NW * 64) {
        const int nn = e / F, col = e % F;
        const float v = wv(nn, col) * s_w;
        const _Float16 vh = (_Float16)v;
        const int pos = ((((col >> 5) * 8 + (nn >> 4)) * 2 + ((nn >> 3) & 1)) * 32 + (col & 31)) * 8 + (nn & 7);
        whi[pos] = vh;
        wlo[pos] = (_Float16)(v - (float)vh);
    }
    __syncthreads();
    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t stride = (int64_t)gridDim.x * NW;
    // G_0 of a tile: k block b = rows 16 b .. + 15 (b < 4: head a, else head b); lane (j, h) takes rows 16 b + 8 h + e of sample j
    auto fetch = [&](int64_t t, float (&v)[8][8]) {
        const float *ga = a.g_a + t * a.gs_a + (8 * h) * 32 + j, *gb = a.g_b + t * a.gs_b + (8 * h) * 32 + j;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const float *src = (b < 4 ? ga : gb) + (16 * (b & 3)) * 32;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[b][e] = src[e * 32];
        }
    };
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    float nxt[PREFETCH ? 8 : 1][8];
    if constexpr (PREFETCH) { if (tile < n_tiles) fetch(tile, nxt); }
#pragma clang loop unroll(disable)
    for (; tile < n_tiles; tile += stride) {
        float mx = 0.0f;
        u32x4h bh[8], bl[8];
        {
            float cur[8][8];
            if constexpr (PREFETCH) {
#pragma unroll
                for (int b = 0; b < 8; ++b)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cur[b][e] = nxt[b][e];
                const int64_t tn_ = tile + stride;
                fetch(tn_ < n_tiles ? tn_ : tile, nxt);                // next tile's rows travel under this tile's products
            } else fetch(tile, cur);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(cur[b][e]));
            float s_g, inv_g;
            f2_scales(f2_xmax(mx), s_g, inv_g);
            mx = inv_g;
#pragma unroll
            for (int b = 0; b < 8; ++b) f2_split8(cur[b], s_g, bh[b], bl[b]);
        }
        const float c = inv_w * mx;
        float *outp = a.out + tile * a.out_stride;
#pragma unroll 2
        for (int ob = 0; ob < NOB; ++ob) {
            unsigned mb = 0xffffffffu;
            if (a.mask != nullptr) mb = a.mask[tile * a.mask_stride + ob * 64 + lane];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const _Float16 *ph = whi + (size_t)(ob * 8) * 512 + lane * 8, *pl = wlo + (size_t)(ob * 8) * 512 + lane * 8;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u32x4h ah = *reinterpret_cast<const u32x4h *>(ph + b * 512), al = *reinterpret_cast<const u32x4h *>(pl + b * 512);
                acc = mfma_f16(al, bh[b], acc);
                acc = mfma_f16(ah, bl[b], acc);
                acc = mfma_f16(ah, bh[b], acc);
            }
            tn::pin16(acc);
            f32x16 res;
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = tn::mlp::mask_keep(acc[r] * c, mb, r);
            tn::layers::wreg_store_block<true>(outp, ob, j, h, res);
        }
    }
}

template <int F>
int launch_dx(const DxArgs &a, int64_t n, hipStream_t s)
{
    constexpr size_t lds_bytes = (size_t)2 * 128 * F * 2 + 128;
    auto kern = heads_dx_f2_kernel<F>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("heads_dx: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    kern<<<dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + TN_DX_WAVES - 1) / TN_DX_WAVES, 256))), dim3(TN_DX_WAVES * 64), lds_bytes, s>>>(a, n);
    return tn::check_launch("heads_dx_f2_kernel");
}

}  // namespace

// internal entry point (mlp_bwd2.hip, bwd_pair_common): out rows [F] (+ relu' bits) = W_a[:, col0_a ..]^T G_a + W_b[:, col0_b ..]^T G_b
extern "C" __attribute__((visibility("hidden"))) int tn_heads_dx_rows(const float *w_a, int ld_a, int col0_a, const float *w_b, int ld_b, int col0_b,
                                                                     const float *g_a, int64_t gs_a, const float *g_b, int64_t gs_b, int F,
                                                                     float *out, int64_t out_stride, const void *mask, int64_t mask_stride,
                                                                     int64_t n, void *stream)
{
    DxArgs a;
    a.w_a = w_a; a.w_b = w_b; a.ld_a = ld_a; a.col0_a = col0_a; a.ld_b = ld_b; a.col0_b = col0_b;
    a.g_a = g_a; a.g_b = g_b; a.gs_a = gs_a; a.gs_b = gs_b; a.out = out; a.out_stride = out_stride;
    a.mask = reinterpret_cast<const unsigned *>(mask); a.mask_stride = mask_stride;
    if (F == 256) return launch_dx<256>(a, n, (hipStream_t)stream);
    if (F == 128) return launch_dx<128>(a, n, (hipStream_t)stream);
    return tn::fail(TN_E_CONFIG, "heads_dx: 128 or 256 x columns");
}
