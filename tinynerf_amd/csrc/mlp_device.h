// Device building blocks of the fused fp32-MFMA MLP (gfx950).
//
// TRANSPOSED FORMULATION.  A wavefront owns 32 samples and evaluates every layer as
//     D[out_feature][sample] = W[out_feature][in_feature] * X[in_feature][sample]
// with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 cycles, 157 TFLOP/s chip peak):
//   A operand = W   : lane l holds W[32*ob + (l&31)][k],  k = 2*step + (l>>5)
//   B operand = X   : lane l holds X[k][sample l&31]
//   D (16 regs)     : lane l, reg r holds D[32*ob + (r&3) + 8*(r>>2) + 4*(l>>5)][sample l&31]
// Because the MFMA reduction index may be visited in any order, step t of a layer is defined to
// consume in-feature  f_h(t) = 8*(t>>2) + 4*h + (t&3)  (h = l>>5).  With that choice
//   * the D registers of one layer (after bias + ReLU, in place) ARE the B operands of the next:
//     reg r of tile kb is step t = 16*kb + r  ->  activations never leave the register file;
//   * the A operand of steps 4g..4g+3 is W[row][8g+4h .. 8g+4h+3], four consecutive floats of a
//     torch-layout weight row -> one ds_read_b128 from the LDS copy per four MFMAs.
// LDS weight rows are padded by 4 floats (stride/4 odd) so the 16-lane groups of ds_read_b128
// hit distinct 16-byte slots.
#pragma once
#include "tn_common.h"

namespace tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator initialised with the bias of out-tile `ob` (D layout)
__device__ __forceinline__ f32x16 bias_tile(const float *bias, int ob, int h) {
    f32x16 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 32 * ob + 8 * q + 4 * h);
        y[4 * q + 0] = b[0]; y[4 * q + 1] = b[1]; y[4 * q + 2] = b[2]; y[4 * q + 3] = b[3];
    }
    return y;
}

// Materialise an MFMA result in VGPRs HERE, in straight-line code.  hipcc (ROCm 7.2) counts the MFMA ->
// v_accvgpr_read wait states along the longest predecessor path only: when a branch sits between the last
// MFMA and the first read of its accumulator, the short path reads a[15] (written in the final pass) too
// early and gets a stale value -- observed as rare wrong values in output register 15 only.  Every chain
// whose result is consumed under a branch is therefore pinned first.
__device__ __forceinline__ void pin16(f32x16 &v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));
}

// one v_max per element: fmaxf() on a value the compiler cannot prove canonical (an MFMA result behind pin16) costs a second,
// canonicalising v_max (32 extra VALU instructions per 64-wide layer and tile -- VALU time adds to MFMA time on a SIMD)
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float y;
        asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(v[r]));
        v[r] = y;
    }
    return v;
}

// One hidden layer H -> H on register-resident activations: x (D layout of the previous layer)
// is replaced by relu(W x + b).  W is [H][stride] floats, rows padded; bias [H].
template <int H>
__device__ __forceinline__ void hidden_layer(const float *__restrict__ W, const float *__restrict__ bias, int stride,
                                             f32x16 (&x)[H / 32], int i, int h)
{
    constexpr int T = H / 32;
    constexpr int NG = 4 * T;               // groups of 4 reduction steps; group g = (kb, q)
    f32x16 y[T];
#pragma unroll
    for (int ob = 0; ob < T; ++ob) y[ob] = bias_tile(bias, ob, h);
    // software pipeline: the A operands of group g+1 are requested before the MFMAs of group g are issued (left to
    // itself the compiler sinks every ds_read to just before its first use: read -> wait -> 2T MFMAs -> read -> ...)
    f32x4 cur[T], nxt[T];
#pragma unroll
    for (int ob = 0; ob < T; ++ob) cur[ob] = *reinterpret_cast<const f32x4 *>(W + (32 * ob + i) * stride + 4 * h);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kb = g >> 2, q = g & 3;
        if (g + 1 < NG) {
#pragma unroll
            for (int ob = 0; ob < T; ++ob)
                nxt[ob] = *reinterpret_cast<const f32x4 *>(W + (32 * ob + i) * stride + 8 * (g + 1) + 4 * h);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ob = 0; ob < T; ++ob) y[ob] = mfma32(cur[ob][u], x[kb][4 * q + u], y[ob]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ob = 0; ob < T; ++ob) cur[ob] = nxt[ob];
    }
#pragma unroll
    for (int ob = 0; ob < T; ++ob) { pin16(y[ob]); x[ob] = relu16(y[ob]); }
}

// out[o] = <W[o][:], x> + b[o] for a few outputs (o < NOUT <= 4) on the VALU: each lane reduces the
// H/2 features it holds, the two half-waves are combined with one cross-half shuffle.
template <int H>
__device__ __forceinline__ float small_out(const float *__restrict__ Wrow, float bias, const f32x16 (&x)[H / 32], int h)
{
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < H / 32; ++kb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(Wrow + 32 * kb + 8 * q + 4 * h);
            p0 = fmaf(w[0], x[kb][4 * q + 0], p0);
            p1 = fmaf(w[1], x[kb][4 * q + 1], p1);
            p0 = fmaf(w[2], x[kb][4 * q + 2], p0);
            p1 = fmaf(w[3], x[kb][4 * q + 3], p1);
        }
    }
    float p = p0 + p1;
    p += __shfl_xor(p, 32, 64);
    return p + bias;
}

__device__ __forceinline__ float apply_act(float y, int act) {
    if (act == TN_ACT_EXP_M1) return expf(y - 1.0f);                 // models.py:74, truncated_exp fwd = exp
    if (act == TN_ACT_SIGMOID) return 1.0f / (1.0f + expf(-y));      // models.py:85
    return y;
}

// positional-encoding value of flat index p for one point (models.py:36-39):
// p = c*2F + (f | F+f)  ->  sin / cos (x_c * freq_f)
__device__ __forceinline__ float posenc_value(const float *xc, int p, int F, const float *freqs) {
    const int c = p / (2 * F);
    const int rem = p - c * 2 * F;
    const bool is_cos = rem >= F;
    const int f = is_cos ? rem - F : rem;
    const float fr = freqs ? freqs[f] : ldexpf(3.14159274101257324f, f);
    const float ang = xc[c] * fr;
    return is_cos ? cosf(ang) : sinf(ang);
}

}  // namespace tn
