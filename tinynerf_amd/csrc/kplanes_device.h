// K-Planes bilinear plane lookups (reference src/models.py:93-163) for channel-last planes.
//
// Layout: plane[s][p] is fp32 [H][W][C] -- one texel's C channels are one contiguous 4*C-byte
// run (128 B = one cache line for C = 32), where the reference's [1,C,H,W] layout spreads a texel
// over C cache lines (SURVEY 7 "K-Planes gather locality").  torch sees the same memory as a
// channels_last [1,C,H,W] parameter, so state_dict shapes do not change.
//
// Lane mapping (shared with the fused MLP, mlp_device.h): lane l = (h = l>>5, j = l&31) owns
// sample j of the wave's 32-sample tile and the channel half [h*C/2, (h+1)*C/2) of every plane, read
// as float4: lanes j and j+32 together consume one full texel line.
#pragma once
#include "tn_common.h"

namespace tn {

typedef float f32x4k __attribute__((ext_vector_type(4)));

struct PlaneTaps {
    int off[4];      // element offset of texel (y,x) * C for nw, ne, sw, se; -1 when out of bounds
    float w[4];      // bilinear weights
    int cell;        // id of the (x0,y0) cell the point falls in (also defined out of bounds)
};

// ATen grid_sampler_2d conventions: align_corners=True, zeros padding; u indexes W, v indexes H.
__device__ __forceinline__ PlaneTaps plane_taps(float u, float v, int H, int W, int C) {
    // One rounding per operation, like ATen, and the same bits in every kernel that inlines this: under -ffp-contract=fast the
    // compiler folds the product ix = t * (W - 1) into the subtractions that follow (fx = fma(t, W - 1, -x0)) wherever it
    // sees fit -- observed in one kernel and not in another: features 6e-6 apart at W = 96 -- per-statement pragmas
    // notwithstanding; the empty asm statements pin the sums and the products.
    PlaneTaps t;
    float u1 = u + 1.0f, v1 = v + 1.0f;
    asm volatile("" : "+v"(u1), "+v"(v1));
    float ix = (u1 * 0.5f) * (float)(W - 1);
    float iy = (v1 * 0.5f) * (float)(H - 1);
    asm volatile("" : "+v"(ix), "+v"(iy));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const float gx = (x0f + 1.0f) - ix, gy = (y0f + 1.0f) - iy;
    t.w[0] = gx * gy; t.w[1] = fx * gy; t.w[2] = gx * fy; t.w[3] = fx * fy;
    const bool bx0 = x0f >= 0.0f && x0f < (float)W, bx1 = x0f + 1.0f >= 0.0f && x0f + 1.0f < (float)W;
    const bool by0 = y0f >= 0.0f && y0f < (float)H, by1 = y0f + 1.0f >= 0.0f && y0f + 1.0f < (float)H;
    const int x0 = bx0 ? (int)x0f : 0, x1 = bx1 ? (int)x0f + 1 : 0;
    const int y0 = by0 ? (int)y0f : 0, y1 = by1 ? (int)y0f + 1 : 0;
    t.off[0] = (bx0 && by0) ? (y0 * W + x0) * C : -1;
    t.off[1] = (bx1 && by0) ? (y0 * W + x1) * C : -1;
    t.off[2] = (bx0 && by1) ? (y1 * W + x0) * C : -1;
    t.off[3] = (bx1 && by1) ? (y1 * W + x1) * C : -1;
    const float cx = fminf(fmaxf(x0f, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(y0f, -2.0f), (float)H + 1.0f);
    t.cell = ((int)cy + 2) * (W + 4) + (int)cx + 2;
    return t;
}

// interpolate NV float4 groups (channels c0 .. c0+4*NV) of one plane.  All 4*NV loads are unconditional (an
// out-of-bounds tap reads texel 0 with weight 0): a load under `if (in bounds)` has to be waited for at the end of
// that branch, which serialises the four taps' L2 latencies instead of overlapping all loads of a plane.
// CS = distance in floats between consecutive groups: 4 = one contiguous channel run (lane owns channels c0 .. c0+4*NV),
// 8 = the MFMA B-operand pattern of mlp_device.h (lane half h owns channels 8v + 4h .. +3, v = 0..NV-1, with c0 = 4h)
template <int NV, int CS = 4>
__device__ __forceinline__ void plane_gather(const float *__restrict__ plane, const PlaneTaps &t, int c0, f32x4k (&out)[NV]) {
    // SGPR-base addressing (wave-uniform plane pointer + one 32-bit byte offset per lane + immediate): a 64-bit per-lane
    // address costs the SIMD measurably more issue time per vector-memory instruction (scripts/microbench/wreg_layer.hip)
    f32x4k tex[4][NV];
    float w[4];
    const char *base = reinterpret_cast<const char *>(plane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool in = t.off[k] >= 0;
        unsigned boff = (unsigned)((in ? t.off[k] : 0) + c0) * 4u;         // planes hold at most 2^30 floats
        asm volatile("" : "+v"(boff));                                     // (keeps the zero-extension at the access)
        w[k] = in ? t.w[k] : 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) tex[k][v] = *reinterpret_cast<const f32x4k *>(base + boff + (unsigned)(v * CS * 4));
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        f32x4k acc = tex[0][v] * w[0];          // an explicit fma chain: the same instructions in every kernel
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(tex[k][v][e], w[k], acc[e]);
        out[v] = acc;
    }
}

// coordinate pairs of the three planes of a scale: itertools.combinations(range(3), 2) (models.py:146)
__device__ __forceinline__ void pair_uv(const float x[3], int p, float &u, float &v) {
    u = (p == 2) ? x[1] : x[0];
    v = (p == 0) ? x[1] : x[2];
}

}  // namespace tn
