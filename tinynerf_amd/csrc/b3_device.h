// "bf16x3": fp32 matrix products on the bf16 matrix cores with EXACT three-way operand splits.
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 at 16 x that.  Every fp32 value
// is the exact sum of three bf16 values obtained by truncation,
//     x = hi + mid + lo,   hi = top 8 significant bits, mid = the next 8, lo = the last 8   (x - hi and (x - hi) - mid are exact),
// so  w x = sum of nine bf16 x bf16 products, each exact in fp32.  Six of them are kept -- hh, hm, mh, mm, hl, lh -- the other
// three (ml, lm, ll) are below 2^-23 |w x|, i.e. below half an ulp of the fp32 product itself; the MFMA accumulates in fp32.
// Measured (scripts/microbench/bf16x3_layer.hip): the same error against an fp64 evaluation as the fp32 MFMA (2-4e-7 of the
// largest output of a 64 x 64 layer), 1.6 x its speed for a 64-wide register-resident layer where every operand value is
// split by the wave that multiplies it; the wide-stack kernels split each activation once per workgroup and each weight once
// per launch (6 MFMAs x 32 cycles per 32 x 32 x 16 block against 8 x 64: 2.67 x the matrix rate).
#pragma once
#include "tn_common.h"

namespace tn {
namespace b3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Op { u32x4 hi, mid, lo; };       // 8 values as three packed-bf16 operands (one MFMA A or B operand each)

// two fp32 values -> packed bf16 pairs (low half = first value) of their three terms
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned ha = __float_as_uint(a) & 0xffff0000u, hb = __float_as_uint(b) & 0xffff0000u;
    const float ra = a - __uint_as_float(ha), rb = b - __uint_as_float(hb);            // exact
    const unsigned ma = __float_as_uint(ra) & 0xffff0000u, mb = __float_as_uint(rb) & 0xffff0000u;
    const float la = ra - __uint_as_float(ma), lb = rb - __uint_as_float(mb);          // exact, <= 8 significant bits
    hi = __builtin_amdgcn_perm(hb, ha, 0x07060302u);            // bytes: [ha.2, ha.3, hb.2, hb.3]
    mid = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ Op split8(const float (&v)[8]) {
    Op o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned a, b, c;
        split2(v[2 * p], v[2 * p + 1], a, b, c);
        o.hi[p] = a; o.mid[p] = b; o.lo[p] = c;
    }
    return o;
}

__device__ __forceinline__ f32x16 mfma16(const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc += A B over one 16-wide k block, six partial products, small terms first
__device__ __forceinline__ f32x16 mfma6(const Op &A, const Op &B, f32x16 acc) {
    acc = mfma16(A.lo, B.hi, acc);
    acc = mfma16(A.hi, B.lo, acc);
    acc = mfma16(A.mid, B.mid, acc);
    acc = mfma16(A.mid, B.hi, acc);
    acc = mfma16(A.hi, B.mid, acc);
    acc = mfma16(A.hi, B.hi, acc);
    return acc;
}

}  // namespace b3
}  // namespace tn
