// Kernel arguments, LDS weight staging and first-layer input fetch shared by the fused MLP forward
// (mlp.hip) and backward (mlp_bwd2.hip, mlp_bwd_layers.hip) kernels.
#pragma once
#include "mlp_device.h"

namespace tn {
namespace mlp {

using tn::f32x16;
using tn::f32x4;

struct MlpArgs {
    int n_layers, in_dim, K0, K0_pad, enc, n_freqs, out_act, out_dim;
    const float *freqs;
    const float *W[TN_MLP_MAX_LAYERS];
    const float *B[TN_MLP_MAX_LAYERS];
    int w_off[TN_MLP_MAX_LAYERS], b_off[TN_MLP_MAX_LAYERS], stride[TN_MLP_MAX_LAYERS];
    int K[TN_MLP_MAX_LAYERS], N[TN_MLP_MAX_LAYERS];     // true in / out width of each layer
    int lds_floats;
    int accum_gx;          // grad_x += (TN_MLP_ACCUM_GRAD_X)
    const int *aux_index;  // TN_ENC_AUX_CAT: x row -> aux table row (nullptr: identity)
    int aux_stride;
    const float *row_gate; // forward only: tiles whose gates are all 0 are skipped (output 0)
    // row views (tn_mlp_desc::x_rows / grad_x_rows): x^T and d loss / d x as [feature][32-sample] rows per tile
    const float *x_rows;
    float *gx_rows;
    int64_t x_rows_stride, gx_rows_stride;
    int b3;                // TN_MLP_BF16X3 (or TN_MLP_F16X2, which implies it for the weight gradients)
    int f2;                // TN_MLP_F16X2: forward / data-gradient layers as fp16 two-term splits
    int rows_only;         // TN_MLP_ROWS_ONLY (producer: no row-major y)
    int x_from_rows;       // TN_MLP_X_FROM_ROWS (consumer: x only exists as x_rows)
    int x_wgrad_done = 0;                       // the first layer's x-column weight gradient was taken by the caller (tn_mlp_bwd_pair)
    const unsigned *gx_mask_rows = nullptr;     // tn_mlp_desc::grad_x_mask_rows: relu' bits applied to grad_x_rows
    int64_t gx_mask_stride = 0;
    int skip_last = 0;                          // TN_MLP_SKIP_LAST
    int layerwise = 0;     // TN_MLP_LAYERWISE: no cross-layer persistent launch (mlp_fused_f2.hip)
    int lean;              // TN_MLP_LEAN: the training forward leaves the H rows of the workspace unwritten (mlp_wgrad_rc.hip rebuilds them)
    int f2_plane[TN_MLP_MAX_LAYERS], f2_scale;     // f16x2 heads (mlp_f2_heads.h): halfs per weight plane, float index of the (s, 1 / s) pairs
};

// column of the torch weight matrix that feeds first-layer slot q (slot order: see fetch_input)
__device__ __forceinline__ int layer0_col(const MlpArgs &a, int q) {
    if (a.enc == TN_ENC_DIR_CAT || a.enc == TN_ENC_AUX_CAT) {
        const int pe = a.K0 - a.in_dim;                    // 6F + 3 for the colour head
        return q < a.in_dim ? pe + q : q - a.in_dim;       // torch order: [PE(d), d, feat]
    }
    return q;
}

// copy every layer into LDS: [rows][K_pad + 4] + bias, zero padded
__device__ inline void stage_weights(const MlpArgs &a, float *lds) {
    for (int l = 0; l < a.n_layers; ++l) {
        const int stride = a.stride[l];
        const int rows = (a.b_off[l] - a.w_off[l]) / stride;
        const int K = a.K[l], N = a.N[l];
        float *w = lds + a.w_off[l];
        for (int e = threadIdx.x; e < rows * stride; e += blockDim.x) {
            const int r = e / stride, q = e - r * stride;
            float v = 0.0f;
            if (r < N && q < K) v = a.W[l][(int64_t)r * K + (l == 0 ? layer0_col(a, q) : q)];
            w[e] = v;
        }
        float *b = lds + a.b_off[l];
        const int brow = (rows + 3) & ~3;
        for (int e = threadIdx.x; e < brow; e += blockDim.x) b[e] = e < N ? a.B[l][e] : 0.0f;
    }
}

// The four first-layer inputs of slots 8g+4h .. +3 for this lane's sample.
//   TN_ENC_NONE    slot q = x[q]
//   TN_ENC_POSENC  slot q = PE(x)[q]
//   TN_ENC_DIR_CAT slot q = x[q] (q < in_dim), PE(d)[q-in_dim], d[..], 0 padding
//   TN_ENC_AUX_CAT slot q = x[q] (q < in_dim), auxrow[q-in_dim] (table row, zero padded to K0_pad)
__device__ __forceinline__ f32x4 fetch_input(const MlpArgs &a, const float *__restrict__ xrow, const float *aux3,
                                             bool valid, int g, int h, const float *__restrict__ auxrow = nullptr)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!valid) return v;
    const int q0 = 8 * g + 4 * h;
    if (a.enc == TN_ENC_AUX_CAT)
        return q0 < a.in_dim ? *reinterpret_cast<const f32x4 *>(xrow + q0) : *reinterpret_cast<const f32x4 *>(auxrow + (q0 - a.in_dim));
    if (a.enc == TN_ENC_POSENC) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q0 + u < a.K0) v[u] = tn::posenc_value(aux3, q0 + u, a.n_freqs, a.freqs);
        return v;
    }
    if (q0 + 3 < a.in_dim && (a.in_dim & 3) == 0) return *reinterpret_cast<const f32x4 *>(xrow + q0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u;
        if (q < a.in_dim) v[u] = xrow[q];
        else if (a.enc == TN_ENC_DIR_CAT) {
            const int p = q - a.in_dim;
            if (p < 6 * a.n_freqs) v[u] = tn::posenc_value(aux3, p, a.n_freqs, a.freqs);
            else if (p < 6 * a.n_freqs + 3) v[u] = aux3[p - 6 * a.n_freqs];
        }
    }
    return v;
}

// A operand of 4 consecutive steps: LDS copy (padded, always in range) or guarded global read
template <bool WLDS>
__device__ __forceinline__ f32x4 load_a4(const float *__restrict__ W, int row, int col, int K, int stride) {
    if constexpr (WLDS) {
        return *reinterpret_cast<const f32x4 *>(W + row * stride + col);
    } else {
        if (col + 3 < K && (K & 3) == 0) return *reinterpret_cast<const f32x4 *>(W + (int64_t)row * K + col);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (col + u < K) v[u] = W[(int64_t)row * K + col + u];
        return v;
    }
}

constexpr int LDS_LIMIT_BYTES = 160 * 1024;

// ---- activation workspace of the two-pass backward (mlp_bwd2.hip), also written by the training forward ----
// Rows of [32 samples] floats (128 B) per 32-sample tile:
//   H_1..H_NH | G_0..G_{NH-1} | g_pre (4) | E (encoded first-layer slots of TN_ENC_DIR_CAT / POSENC that are not plain
//   x columns) | pre (4, last layer's pre-activation) | ReLU bit masks (2 rows per (layer, 32-feature block))
// The weight-gradient kernel stages the first stash_rows_w() rows; pre and masks are for the chain kernel only.
__host__ __device__ inline int x_slots(int enc, int in_dim) { return enc == TN_ENC_POSENC ? 0 : in_dim; }
__host__ __device__ inline int extra_rows(int enc, int in_dim, int K0_pad) {
    return (enc == TN_ENC_NONE || enc == TN_ENC_AUX_CAT) ? 0 : K0_pad - x_slots(enc, in_dim);
}
__host__ __device__ inline int stash_rows_w(int H, int nh, int extra) { return 2 * nh * H + 4 + extra; }
__host__ __device__ inline int stash_rows(int H, int nh, int extra) { return stash_rows_w(H, nh, extra) + 4 + 2 * nh * (H / 32); }

// D-layout tile -> workspace rows [feature][32 samples]; two fully used 128-B lines per store instruction
// `rows` is wave-uniform at every call site (a wave owns its tile): the stores take the SGPR-base form -- uniform 64-bit base,
// one 32-bit lane offset, the 16 rows as immediate offsets.  (A 64-bit per-lane address costs the SIMD measurably more issue
// time per vector-memory instruction, time the matrix pipe stands still: scripts/microbench/wreg_layer.hip.)
typedef __attribute__((address_space(1))) char global_char;          // (an integer round trip must not turn the pointer generic: flat_store)
__device__ __forceinline__ global_char *wave_uniform_global(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<global_char *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void store_rows(float *__restrict__ rows, const f32x16 &t, int ob, int j, int h) {
    global_char *base = wave_uniform_global(rows + 32 * ob * 32);
    unsigned off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(off));                          // (keeps the zero-extension at the access)
#pragma unroll
    for (int r = 0; r < 16; ++r)          // (nt hints on these stores: no effect on the heads' launches, measured on all three configurations)
        *reinterpret_cast<__attribute__((address_space(1))) float *>(base + off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128)) = t[r];
}

// the same rows back into the D layout
__device__ __forceinline__ void load_rows(const float *__restrict__ rows, f32x16 &t, int ob, int j, int h) {
    const global_char *base = wave_uniform_global(rows + 32 * ob * 32);
    unsigned off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int r = 0; r < 16; ++r)
        t[r] = *reinterpret_cast<const __attribute__((address_space(1))) float *>(base + off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128));
}

// Bit r = (t[r] > 0) for ReLU outputs (t >= 0, so "> 0" is "bit pattern != 0"; -0.0 cannot occur after fmaxf(x, 0)
// ... it can: fmaxf(-0.0, 0) may return either zero, hence the shift that drops the sign bit).  Two VALU operations per
// element (min, shift-or) instead of compare + select + or: VALU instructions are what the MFMA-heavy kernels run out of.
__device__ __forceinline__ unsigned relu_bits(const f32x16 &t) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= min(__float_as_uint(t[r]) << 1, 1u) << r;
    return m;
}

// x where bit r of mask is set, else 0: sign-extend the bit to a full word (v_bfe_i32) and AND it in -- two VALU
// operations instead of and + compare + select
__device__ __forceinline__ float mask_keep(float x, unsigned mask, int r) {
    const int keep = __builtin_amdgcn_sbfe((int)mask, r, 1);
    return __uint_as_float(__float_as_uint(x) & (unsigned)keep);
}

// Configurations of the two-pass backward (chain + weight-gradient kernels, mlp_bwd2.hip) and of the register-resident
// training forward that writes its workspace (mlp.hip): width-64 heads with one or four hidden layers and <= 4 outputs -- the
// reference's sigma / colour decoders.  Everything else (Vanilla 256 x 9, Cobafa 128 x 6, odd shapes) takes the
// layer-by-layer form (mlp_bwd_layers.hip).
inline bool two_pass_supported(const tn_mlp_desc *d) {
    if (!d) return false;
    const int L = d->n_layers, H = d->dims[1];
    if ((L != 2 && L != 5) || H != 64) return false;
    for (int l = 1; l < L; ++l) if (d->dims[l] != H) return false;
    if (d->dims[L] < 1 || d->dims[L] > 4) return false;
    const int K0_pad = (d->dims[0] + 7) & ~7;
    const int T = H / 32, Tk0 = (K0_pad + 31) / 32;
    if (T * Tk0 + (L - 2) * T * T + T > 48) return false;       // 16 waves x 3 accumulator tiles
    const int extra = extra_rows(d->encoding, d->in_dim, K0_pad);
    const int xs = x_slots(d->encoding, d->in_dim);
    if (xs > 0 && (d->in_dim & 3)) return false;
    const int R = stash_rows_w(H, L - 1, extra);
    const int aw = d->encoding == TN_ENC_AUX_CAT ? K0_pad - d->in_dim : 0;
    if (R * 8 + (xs > 0 ? 8 * d->in_dim : 0) > 10 * 1024) return false;      // prefetch registers of the wgrad kernel
    if (((size_t)R * 36 + (xs > 0 ? 32 * (size_t)d->in_dim : 0) + 32 * (size_t)aw) * 4 > 160 * 1024 || aw > 64) return false;
    return true;
}

inline int plan(const tn_mlp_desc *d, MlpArgs &a, int &H)
{
    TN_REQUIRE(d, TN_E_NULL, "mlp: null descriptor");
    const int L = d->n_layers;
    TN_REQUIRE(L >= 2 && L <= TN_MLP_MAX_LAYERS, TN_E_CONFIG, "mlp: n_layers must be in [2, 12]");
    H = d->dims[1];
    TN_REQUIRE(H == 32 || H == 64 || H == 128 || H == 256, TN_E_CONFIG, "mlp: hidden width must be 32, 64, 128 or 256");
    for (int l = 1; l < L; ++l) TN_REQUIRE(d->dims[l] == H, TN_E_CONFIG, "mlp: all hidden layers must share one width");
    for (int l = 0; l < L; ++l) TN_REQUIRE(d->weights[l] && d->biases[l], TN_E_NULL, "mlp: null weight / bias pointer");
    a.n_layers = L; a.in_dim = d->in_dim; a.K0 = d->dims[0]; a.K0_pad = (a.K0 + 7) & ~7;
    a.enc = d->encoding; a.n_freqs = d->n_freqs; a.out_act = d->out_activation; a.out_dim = d->dims[L]; a.accum_gx = d->flags & TN_MLP_ACCUM_GRAD_X;
    a.freqs = d->freqs; a.aux_index = d->aux_index; a.aux_stride = d->aux_stride; a.row_gate = d->row_gate;
    a.b3 = (d->flags & (TN_MLP_BF16X3 | TN_MLP_F16X2)) != 0;
    a.f2 = (d->flags & TN_MLP_F16X2) != 0;
    a.rows_only = (d->flags & TN_MLP_ROWS_ONLY) != 0;
    a.x_from_rows = (d->flags & TN_MLP_X_FROM_ROWS) != 0;
    a.lean = (d->flags & TN_MLP_LEAN) != 0;
    a.skip_last = (d->flags & TN_MLP_SKIP_LAST) != 0;
    a.layerwise = (d->flags & TN_MLP_LAYERWISE) != 0;
    a.gx_mask_rows = reinterpret_cast<const unsigned *>(d->grad_x_mask_rows); a.gx_mask_stride = d->grad_x_mask_tile_stride;
    TN_REQUIRE(!a.gx_mask_rows || (d->grad_x_rows && (((uintptr_t)a.gx_mask_rows) & 3) == 0), TN_E_CONFIG, "mlp: grad_x_mask_rows goes with grad_x_rows");
    a.x_rows = d->x_rows; a.gx_rows = d->grad_x_rows; a.x_rows_stride = d->x_rows_tile_stride; a.gx_rows_stride = d->grad_x_rows_tile_stride;
    TN_REQUIRE((!a.x_rows && !a.gx_rows) || ((a.in_dim & 31) == 0 && a.enc != TN_ENC_POSENC), TN_E_CONFIG,
               "mlp: x_rows / grad_x_rows need in_dim % 32 == 0 and an encoding that keeps x as input columns");
    TN_REQUIRE((!a.x_rows || (((uintptr_t)a.x_rows & 15) == 0 && a.x_rows_stride >= 32 * (int64_t)a.in_dim && (a.x_rows_stride & 3) == 0)) &&
                   (!a.gx_rows || (((uintptr_t)a.gx_rows & 15) == 0 && a.gx_rows_stride >= 32 * (int64_t)a.in_dim)), TN_E_ALIGN,
               "mlp: x_rows / grad_x_rows must be 16-byte aligned with a tile stride >= 32 * in_dim floats");
    TN_REQUIRE(a.out_dim >= 1 && a.in_dim >= 1, TN_E_SIZE, "mlp: bad in/out width");
    switch (a.enc) {
    case TN_ENC_NONE: TN_REQUIRE(a.K0 == a.in_dim, TN_E_CONFIG, "mlp: dims[0] must equal in_dim"); break;
    case TN_ENC_POSENC:
        TN_REQUIRE(a.in_dim == 3 && a.K0 == 6 * a.n_freqs, TN_E_CONFIG, "mlp: posenc expects in_dim 3 and dims[0] = 6F");
        break;
    case TN_ENC_DIR_CAT:
        TN_REQUIRE(a.K0 == a.in_dim + 6 * a.n_freqs + 3, TN_E_CONFIG, "mlp: dir_cat expects dims[0] = in_dim + 6F + 3");
        break;
    case TN_ENC_AUX_CAT:
        TN_REQUIRE(a.K0 > a.in_dim && (a.in_dim & 3) == 0 && (a.aux_stride & 3) == 0 && a.aux_stride >= a.K0_pad - a.in_dim, TN_E_CONFIG,
                   "mlp: aux_cat expects in_dim % 4 == 0 and aux_stride % 4 == 0, aux_stride >= pad8(dims[0]) - in_dim");
        break;
    default: return tn::fail(TN_E_CONFIG, "mlp: unknown encoding");
    }
    int off = 0;
    for (int l = 0; l < L; ++l) {
        a.W[l] = d->weights[l]; a.B[l] = d->biases[l];
        a.K[l] = d->dims[l]; a.N[l] = d->dims[l + 1];
        const int Kp = l == 0 ? a.K0_pad : H;
        const int rows = (l == L - 1) ? (a.out_dim <= 4 ? a.out_dim : ((a.out_dim + 31) & ~31)) : H;
        a.stride[l] = Kp + 4;
        a.w_off[l] = off; off += rows * a.stride[l];
        a.b_off[l] = off; off += (rows + 3) & ~3;
    }
    a.lds_floats = off;
    return TN_OK;
}


}  // namespace mlp
}  // namespace tn
