// Shared pieces of the layer-by-layer kernels of the wide stacks (mlp_bwd_layers.hip: exact fp32 MFMA; mlp_b3_layers.hip: the
// same layers on bf16 MFMA with exact three-way operand splits): argument structs, the [feature][32-sample] row helpers.
#pragma once
#include "mlp_stage.h"

namespace tn {
namespace layers {

using tn::f32x16;
using tn::f32x4;

__device__ __forceinline__ int frow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }     // D-layout row of reg r

// data gradient of one layer:  Gout = relu'(Hmask) * (W^T Gin)      (FIRST: grad_x, no mask, row-major out)
struct DgradArgs {
    const float *W;       // [N][K] torch layout
    int N, K;             // rows / columns of W
    int rows_total;       // stash rows per tile: row set at offset `off` of tile t = rows [t rows_total + off, ...) of the workspace
    int64_t off_gin, off_gout, off_mask;    // row offsets (tile-major layout: inside a tile; slab layout: slab base + offset inside the slab's tile, see RowMap)
    int64_t off_bits;                   // >= 0: ReLU bit rows of the mask activation (dgrad_wreg_kernel), else float mask rows
    int enc, in_dim, n_freqs;           // FIRST only: column permutation of layer 0
    int accum_gx;                       // FIRST only: grad_x += (TN_MLP_ACCUM_GRAD_X)
    float *max_in = nullptr;            // f16x2: receives the largest |value| of the rows at off_gin (atomic max; scale of the layer's weight gradient)
};

// forward of one layer on workspace rows
struct FwdLayerArgs {
    const float *W, *B;   // [N][K] torch layout, [N]
    int N, K;
    int Kp;               // input rows present in the workspace (K for hidden layers, K0_pad for the encoded first layer)
    int rows_total;
    int64_t off_in, off_out;
    int out_act;
    int64_t off_bits;     // >= 0: the output activation's ReLU bits go to these rows (2 per 32-feature block), < 0: not wanted
    float *max_in = nullptr;            // f16x2: receives the largest |value| of the rows at off_in (see DgradArgs)
};

// weight gradient of one layer: dW[N][K] += G[N][s] A[K][s]^T over all samples; db[N] += sum_s G
struct WgradArgs {
    float *gW, *gB;
    int N, K, K_pad;
    int rows_total;
    int64_t off_g, off_a, off_e;
    int first, enc, in_dim, n_freqs, xs;
    const float *g_max = nullptr, *a_max = nullptr;       // f16x2: largest |value| of the G rows / of the A rows over ALL tiles
};

// Vector-memory instructions of these kernels use the SGPR-base form (wave-uniform 64-bit base + one 32-bit lane offset +
// immediate): a 64-bit per-lane address costs the SIMD measurably more matrix-pipe time per instruction
// (scripts/microbench/wreg_layer.hip: 84.8 -> 87.8 % busy for the same loads and stores).
__device__ __forceinline__ const float *urow(const float *base, int64_t row) {          // wave-uniform row pointer
    const int64_t o = row * 32;
    return base + (((int64_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)o));
}
__device__ __forceinline__ float *urow(float *base, int64_t row) { return const_cast<float *>(urow(const_cast<const float *>(base), row)); }

// the 32 rows [32 ob, 32 ob + 32) of a [row][32 samples] tile -> registers (lane (j, h): rows 32 ob + 16 h + 0..15, sample j)
__device__ __forceinline__ void wreg_load_rows(const float *__restrict__ rows, int ob, int j, int h, float (&stage)[16]) {
    const char *p = reinterpret_cast<const char *>(rows + 32 * ob * 32);
    unsigned off = (unsigned)(16 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(off));       // (keeps the zero-extension next to the access: base + zext(off) selects the SGPR-base form)
#pragma unroll
    for (int e = 0; e < 16; ++e) stage[e] = *reinterpret_cast<const float *>(p + off + (unsigned)(e * 128));
}
// D-layout rows of block `ob` (lane (j, h), reg r: row 32 ob + frow(r, h), sample j) from / to [row][32 samples] rows
// (NT: non-temporal stores -- rows that the next launch streams once and nobody re-reads from a cache)
template <bool NT = false>
__device__ __forceinline__ void wreg_store_block(float *__restrict__ rows, int ob, int j, int h, const f32x16 &v) {
    char *p = reinterpret_cast<char *>(rows + 32 * ob * 32);
    unsigned off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float *d = reinterpret_cast<float *>(p + off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128));
        if constexpr (NT) __builtin_nontemporal_store(v[r], d); else *d = v[r];
    }
}
__device__ __forceinline__ void wreg_load_block(const float *__restrict__ rows, int ob, int j, int h, float (&m)[16]) {
    const char *p = reinterpret_cast<const char *>(rows + 32 * ob * 32);
    unsigned off = (unsigned)(4 * h * 32 + j) * 4u;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] = *reinterpret_cast<const float *>(p + off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128));
}
// ... -> LDS tile [sample j][feature]: four ds_write_b128
__device__ __forceinline__ void wreg_write_rows(float *__restrict__ tile, int SW, int ob, int j, int h, const float (&stage)[16]) {
    float *p = tile + j * SW + 32 * ob + 16 * h;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        *reinterpret_cast<f32x4 *>(p + 4 * v) = f32x4{stage[4 * v], stage[4 * v + 1], stage[4 * v + 2], stage[4 * v + 3]};
}


// ---- bf16x3 forms (mlp_b3_layers.hip); same arguments, same workspace layout, results equal to fp32 rounding ----
int launch_fwd_b3(int H, bool last, const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s);
int launch_dgrad_b3(int H, const DgradArgs &d, int64_t n, float *stash, hipStream_t s);
int launch_wgrad_b3(int H, const WgradArgs &w, int64_t n, const float *stash, hipStream_t s);
// ---- f16x2 forms of the forward / data gradient (mlp_f2_layers.hip): two-term fp16 splits with power-of-two scales ----
int launch_fwd_f2(int H, bool last, const FwdLayerArgs &f, int64_t n, float *stash, float *y, hipStream_t s);
int launch_fwd_first_f2(int H, const FwdLayerArgs &f, int64_t n, float *stash, hipStream_t s);
int launch_dgrad_f2(int H, const DgradArgs &d, int64_t n, float *stash, hipStream_t s);
int launch_wgrad_f2(int H, const WgradArgs &w, int64_t n, const float *stash, hipStream_t s);

// ---- all layers of a wide stack in one persistent launch (mlp_fused_f2.hip, round 6): inference forward and training forward ----
struct FusedStash {                 // training forward: where the rows the layer-wise backward reads go (RowMap, slab layout)
    float *rows;                    // workspace base
    int rows_total;
    int64_t off_e;                  // the encoded input rows (written by enc_rows_kernel before the launch)
    int n_run;                      // layers to evaluate: n_layers, or n_layers - 1 under TN_MLP_SKIP_LAST
    int64_t off_out[TN_MLP_MAX_LAYERS], off_bits[TN_MLP_MAX_LAYERS];
    float *tail;                    // tail[l]: largest |input value| of layer l (l >= 1), zeroed by the caller
};
struct FusedChain {                 // data-gradient chain: chain position i = layer top - i
    float *rows; int rows_total;
    int64_t off_in;                 // the gradient it starts from
    int64_t off_out[TN_MLP_MAX_LAYERS], off_bits[TN_MLP_MAX_LAYERS];
    int tail_idx[TN_MLP_MAX_LAYERS];
    float *tail;
};
int launch_fused_chain_f2(int H, const tn::mlp::MlpArgs &a, int top, int64_t n, void *pack_area, hipStream_t s, const FusedChain *spec);
int64_t fused_pack_bytes(int H, int L);
bool fused_fwd_ok(int H, const tn::mlp::MlpArgs &a);
int launch_fused_fwd_f2(int H, const tn::mlp::MlpArgs &a, int64_t n, const float *e_rows, float *y, void *pack_area, hipStream_t s, const FusedStash *spec = nullptr);

}  // namespace layers
}  // namespace tn
