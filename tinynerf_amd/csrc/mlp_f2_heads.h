// Width-64 heads (reference models.py:70-89: VanillaOpacityDecoder / VanillaColorDecoder) on the fp16 matrix cores: the f16x2
// arithmetic of mlp_f2_layers.hip -- two-term fp16 splits of both operands, three partial products, power-of-two scales taken
// out of the fp32 accumulators again -- inside the register-resident forward of mlp.hip (TN_MLP_F16X2 on a head whose first
// layer takes plain columns: TN_ENC_NONE / TN_ENC_AUX_CAT, in_dim % 16 == 0, <= 4 outputs, weights in LDS).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate: 64 matrix-pipe cycles per 2 k values -- 4096 cycles per 64 x 64 layer and
// 32-sample tile.  v_mfma_f32_32x32x16_f16 takes 32 cycles per 16 k values: three of them per k block = 768 cycles, and since
// VALU time ADDS to matrix time on a SIMD the ~170 instructions of converting a layer's 32 activations per lane
// still leave the layer ~2.5 x faster.  What makes it fit where bf16x3 did not: two fp16 planes are 4 bytes per weight, exactly
// the fp32 copy's LDS footprint (bf16 triplets: 6).
//
//   * A operand = weights, staged once per workgroup as hi / lo planes [row][k block][half h][8] -- the 8 values lane (row, h)
//     feeds to the MFMA of a k block contiguous (one ds_read_b128 per plane) -- scaled by 2^(9 - e) of the layer's largest
//     |weight|; the last layer (<= 4 outputs: a VALU dot product) keeps its fp32 rows.
//   * B operand = activations in registers.  The D registers of a layer ARE the next layer's inputs as in mlp_device.h: k block b of
//     tile kb = b / 2 is registers 8 (b & 1) .. + 7 -- features 16 b + 4 h + {0..3}, 16 b + 8 + 4 h + {0..3}, the order the planes
//     are staged in.  One scale per SAMPLE and layer from the column maximum: the lane's own 32 values and one exchange with the
//     other half-wave (a wave owns its 32 samples: no cross-wave traffic).
//   * first layer: the x columns (K-Planes: the gathered features, converted once for both heads) and the per-ray table columns
//     of TN_ENC_AUX_CAT accumulate separately with their own scales (either may dwarf the other).
#pragma once
#include "mlp_stage.h"

namespace tn {
namespace mlp {

typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_f16(const u32x4h &a, const u32x4h &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h, a), __builtin_bit_cast(f16x8h, b), c, 0, 0, 0);
}
// 2^(9 - e) and its inverse for a maximum m >= 0 with exponent e (m = 0 / denormal: the largest scale; 0 s = 0)
__device__ __forceinline__ void f2_scales(float m, float &s, float &inv) {
    int se = 263 - (int)(__float_as_uint(m) >> 23);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}
// eight fp32 values (already in k-block order) x s -> the two packed-fp16 operands
__device__ __forceinline__ void f2_split8(const float (&v)[8], float s, u32x4h &hi, u32x4h &lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = v[2 * p] * s, b = v[2 * p + 1] * s;
        const f16x2h hh = {(_Float16)a, (_Float16)b};
        const f16x2h ll = {(_Float16)(a - (float)hh[0]), (_Float16)(b - (float)hh[1])};
        hi[p] = __builtin_bit_cast(unsigned, hh);
        lo[p] = __builtin_bit_cast(unsigned, ll);
    }
}
__device__ __forceinline__ float f2_xmax(float m) { return fmaxf(m, __shfl_xor(m, 32, 64)); }      // both halves of a sample's column

// position of first-layer / hidden slot q inside a staged row: [k block][half][8]
__host__ __device__ __forceinline__ int f2_perm(int q) {
    const int w = q & 15;
    return (q & ~15) + ((w >> 2) & 1) * 8 + (w & 3) + ((w >> 3) << 2);
}

// host: the LDS layout of a head in f16x2 form (MlpArgs::w_off / b_off in floats, stride in halfs for the plane layers)
__host__ inline bool f2_head_ok(const MlpArgs &a, int H) {
    return a.f2 && H == 64 && a.out_dim <= 4 && a.n_layers >= 2 && (a.in_dim & 15) == 0 && (a.enc == TN_ENC_NONE || a.enc == TN_ENC_AUX_CAT) &&
           (a.enc != TN_ENC_AUX_CAT || a.K0_pad - a.in_dim <= 64);
}
__host__ inline void plan_f2(MlpArgs &a, int H) {
    const int L = a.n_layers;
    int off = 0;
    for (int l = 0; l + 1 < L; ++l) {
        const int Kp = ((l == 0 ? a.K0_pad : H) + 15) & ~15;
        a.stride[l] = Kp + 8;                       // halfs per row: (Kp + 8) * 2 B = an odd multiple of 16 B
        a.f2_plane[l] = H * a.stride[l];
        a.w_off[l] = off; off += a.f2_plane[l];     // two planes x 2 B = 4 B per element: `plane` floats
        a.b_off[l] = off; off += H;
    }
    a.stride[L - 1] = H + 4;
    a.w_off[L - 1] = off; off += a.out_dim * (H + 4);
    a.b_off[L - 1] = off; off += 4;
    a.f2_scale = off; off += 2 * L + 32;            // (s, 1 / s) per layer + reduction scratch
    a.lds_floats = (off + 3) & ~3;
}

// device: stage one head (all threads of the workgroup; ends with a barrier)
__device__ inline void stage_weights_f2(const MlpArgs &a, float *lds) {
    const int L = a.n_layers, H = a.N[0];
    float *scales = lds + a.f2_scale, *scratch = scales + 2 * L;
    const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int l = 0; l + 1 < L; ++l) {
        const int K = a.K[l], N = a.N[l], Kp = a.stride[l] - 8;
        float m = 0.0f;
        for (int e = threadIdx.x; e < N * K; e += blockDim.x) m = fmaxf(m, fabsf(a.W[l][e]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __syncthreads();                                    // (scratch of the previous layer has been read)
        if (lane == 0) scratch[wave] = m;
        __syncthreads();
        float g = 0.0f;
        for (int w = 0; w < nw; ++w) g = fmaxf(g, scratch[w]);
        float s, inv;
        f2_scales(g, s, inv);
        if (threadIdx.x == 0) { scales[2 * l] = s; scales[2 * l + 1] = inv; }
        _Float16 *hi = reinterpret_cast<_Float16 *>(lds + a.w_off[l]), *lo = hi + a.f2_plane[l];
        for (int e = threadIdx.x; e < H * Kp; e += blockDim.x) {
            const int r = e / Kp, q = e - r * Kp;
            float v = 0.0f;
            if (r < N && q < K) v = a.W[l][(int64_t)r * K + (l == 0 ? layer0_col(a, q) : q)] * s;
            const _Float16 vh = (_Float16)v;
            const int pos = r * a.stride[l] + f2_perm(q);
            hi[pos] = vh;
            lo[pos] = (_Float16)(v - (float)vh);
        }
        for (int e = threadIdx.x; e < H; e += blockDim.x) lds[a.b_off[l] + e] = e < N ? a.B[l][e] : 0.0f;
    }
    {
        const int l = L - 1, K = a.K[l], N = a.N[l], st = a.stride[l];
        for (int e = threadIdx.x; e < N * st; e += blockDim.x) {
            const int r = e / st, q = e - r * st;
            lds[a.w_off[l] + e] = q < K ? a.W[l][(int64_t)r * K + q] : 0.0f;
        }
        for (int e = threadIdx.x; e < 4; e += blockDim.x) lds[a.b_off[l] + e] = e < N ? a.B[l][e] : 0.0f;
    }
    __syncthreads();
}

// acc[ob] += (W s_W)[32 ob .. + 31][k block b] (x s)[k block b] for both row blocks: three partial products, small terms first
__device__ __forceinline__ void f2_block(const _Float16 *__restrict__ Wh, int plane, int stride, int j, int h, int b, const u32x4h &bh,
                                         const u32x4h &bl, f32x16 (&acc)[2]) {
    const _Float16 *p0 = Wh + j * stride + 16 * b + 8 * h, *p1 = p0 + 32 * stride;
    const u32x4h a0h = *reinterpret_cast<const u32x4h *>(p0), a0l = *reinterpret_cast<const u32x4h *>(p0 + plane);
    const u32x4h a1h = *reinterpret_cast<const u32x4h *>(p1), a1l = *reinterpret_cast<const u32x4h *>(p1 + plane);
    acc[0] = mfma_f16(a0l, bh, acc[0]);
    acc[1] = mfma_f16(a1l, bh, acc[1]);
    acc[0] = mfma_f16(a0h, bl, acc[0]);
    acc[1] = mfma_f16(a1h, bl, acc[1]);
    acc[0] = mfma_f16(a0h, bh, acc[0]);
    acc[1] = mfma_f16(a1h, bh, acc[1]);
}

// one hidden layer 64 -> 64 on register-resident activations (x >= 0: ReLU outputs): x <- relu(W x + b)
__device__ __forceinline__ void hidden_layer_f2(const float *__restrict__ ldsw, const MlpArgs &a, int l, f32x16 (&x)[2], int j, int h) {
    const _Float16 *Wh = reinterpret_cast<const _Float16 *>(ldsw + a.w_off[l]);
    const int plane = a.f2_plane[l], stride = a.stride[l];
    float m = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, x[kb][r]);
    float s, inv;
    f2_scales(f2_xmax(m), s, inv);
    f32x16 y[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[ob][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[b >> 1][8 * (b & 1) + e];
        u32x4h bh, bl;
        f2_split8(v, s, bh, bl);
        f2_block(Wh, plane, stride, j, h, b, bh, bl, y);
        __builtin_amdgcn_sched_barrier(0);
    }
    const float c = inv * ldsw[a.f2_scale + 2 * l + 1];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        pin16(y[ob]);
        const f32x16 bias = bias_tile(ldsw + a.b_off[l], ob, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[ob][r] = fmaf(y[ob][r], c, bias[r]);
        x[ob] = relu16(y[ob]);
    }
}

}  // namespace mlp
}  // namespace tn
