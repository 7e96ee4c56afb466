// Fused MLP heads (reference src/models.py:7-89) as ONE persistent launch on fp32 MFMA.
//
//   y = act_out( W_L relu( ... relu(W_0 enc(x) + b_0) ... ) + b_L )
//
// Replaces the reference's per-layer sgemm + bias + ReLU launches, whose activations round-trip
// HBM between layers.  Here a wavefront owns 32 samples; layer inputs/outputs
// stay in the register file (see mlp_device.h for the transposed MFMA formulation), all layer
// weights are staged once per workgroup into LDS (rows padded, first-layer columns permuted for
// the fused encodings) and re-used by every tile the persistent workgroup processes.  When the
// weights do not fit the 160 KiB LDS (width-256 stacks) the A operands are fetched from L2.
//
// Fused encodings (computed in registers, never materialised):
//   TN_ENC_POSENC  : enc(x) = PE_F(x)                       (models.py:59-68, VanillaFeatureMLP)
//   TN_ENC_DIR_CAT : enc(x, d) = cat[PE_F(d), d, x]          (models.py:79-89, VanillaColorDecoder)
// Output activations: exp(y-1) (models.py:74) and sigmoid (models.py:85).
#include "mlp_stage.h"
#include "mlp_f2_heads.h"
#include "kplanes_device.h"
#include <algorithm>

#ifdef TN_PHASE_TIMERS
__device__ unsigned long long tn_phase_cycles[16];
#define TN_PT_BEGIN unsigned long long pt_ = __builtin_amdgcn_s_memtime();
#define TN_PTG_BEGIN unsigned long long ptg_ = __builtin_amdgcn_s_memtime();
#define TN_PTG(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&tn_phase_cycles[k], n_ - ptg_); ptg_ = n_; }
#define TN_PT(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&tn_phase_cycles[k], n_ - pt_); pt_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int tn_debug_phase_cycles(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(tn_phase_cycles), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(tn_phase_cycles), z, sizeof(z)); }
    return 0;
}
#else
#define TN_PT_BEGIN
#define TN_PT(k)
#define TN_PTG_BEGIN
#define TN_PTG(k)
#endif
namespace {

using tn::f32x16;
using tn::f32x4;
using namespace tn::mlp;

// STASH (training forward, out <= 4, weights in LDS): hidden activations, their ReLU bit masks and the last layer's
// pre-activation also go to the workspace of the two-pass backward (layout: mlp_stage.h), which then starts at the
// output gradient instead of recomputing the forward.
// PAIR (with STASH): a second head on the same x rows is evaluated right behind the first one for every tile (its
// weights sit behind the first head's in LDS), so x is read from HBM once for both.
struct FwdPair { MlpArgs b; const float *aux; float *y; float *stash; };

// KP (with STASH, PAIR, FAST; north star: "the K-Planes bilinear grid sample ... fused into the same launch"): the x rows are
// not read from memory but gathered here -- 3 scales x 3 planes of 32 channels, Hadamard product per scale (models.py:153-163) --
// straight into the first layer's B-operand registers: lane (sample j, half h) gathers channels 8q + 4h .. +3 (q = 0..3) of
// every texel, which are exactly the slots 8g + 4h .. +3 it feeds to the MFMAs of group g = 4 * scale + q.  The feature row
// is also written out once (the backward's weight-gradient and scatter kernels read it); it is never read back here.  While
// one wave waits for its 36 texel lines the other waves of the SIMD run their MFMA layers.
struct KpFwd {
    int H[3], W[3];
    const float *planes[3][3];
    const float *coords;
    int64_t coord_stride;
    float *feat;
    int write_feat = 1;            // 0: inference pair (nobody reads the rows; `feat` is then only a valid dummy address)
};

// FAST: every head takes the plain-column first layer (TN_ENC_NONE with in_dim % 4 == 0, or TN_ENC_AUX_CAT).  Compiled
// separately because the generic first layer drags the sin / cos range reduction of the fused encodings into the kernel
// (14 k VALU instructions, ~120 KB of code against a 64 KB instruction cache shared by two CUs).
// F2 (with FAST, WLDS, H = 64, <= 4 outputs): the layers on the fp16 matrix cores as two-term splits with power-of-two scales
// (mlp_f2_heads.h) -- same inputs, same outputs, same workspace rows.
#ifndef TN_F2_KP_WAVES
#define TN_F2_KP_WAVES 8
#endif
// XB (F2 without KP): the first layer's 16 XB x columns of a sample stay in registers between the column maximum and the products
// (in_dim == 16 XB): ONE round of loads per tile instead of two dependent passes over the row -- at 256 inputs and 12 waves per
// CU the two passes were the launch (0.56 ms for 0.55 M samples: ~32 exposed L2 round trips per tile and wave).
// LEAN (with F2, STASH, PAIR; TN_MLP_LEAN): the hidden activations are NOT written to the workspace -- masks, the last pre-activation and
// the feature rows are; the weight-gradient kernels rebuild H from the feature rows (mlp_wgrad_rc.hip).
template <int H, bool WLDS, int WPB, bool STASH, bool PAIR = false, bool FAST = false, bool KP = false, bool F2 = false, int XB = 0, bool LEAN = false>
__global__ __launch_bounds__(WPB * 64) void mlp_fwd_kernel(MlpArgs a0, const float *__restrict__ x,
                                                           const float *__restrict__ aux0, int64_t n,
                                                           float *__restrict__ y0, float *__restrict__ pre_act0,
                                                           float *__restrict__ stash0, FwdPair pr, KpFwd kp = KpFwd())
{
    static_assert(!KP || (FAST && H == 64), "the fused gather feeds the plain-column first layer of the width-64 heads");
    static_assert(!F2 || (FAST && WLDS && H == 64), "f16x2 heads: plain-column first layer, weights in LDS, width 64");
    static_assert(!LEAN || (F2 && STASH && PAIR), "TN_MLP_LEAN: the paired f16x2 training forward");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = H / 32;
    if constexpr (F2) {
        stage_weights_f2(a0, lds);
        if constexpr (PAIR) stage_weights_f2(pr.b, lds + a0.lds_floats);
    } else if constexpr (WLDS) {
        stage_weights(a0, lds);
        if constexpr (PAIR) stage_weights(pr.b, lds + a0.lds_floats);
        __syncthreads();
    }
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: tile indices and workspace bases are scalars
    const int64_t n_tiles = (n + 31) >> 5;

    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
      // F2: the per-ray table columns of the TN_ENC_AUX_CAT head (ray index -> table row: two dependent round trips) are requested
      // before the gather / the x columns instead of in the middle of the first layer, where nothing hides them
      constexpr bool PREF = F2 && (STASH || WPB <= 8);      // (the 12- / 16-wave inference forms have no registers to spare)
      f32x4 avp[PREF ? 8 : 1];
      const bool pref_second = PAIR && pr.b.enc == TN_ENC_AUX_CAT;
      if constexpr (PREF) {
          const MlpArgs &ax = pref_second ? pr.b : a0;
          if (ax.enc == TN_ENC_AUX_CAT) {                       // (wave-uniform)
              const float *auxp = pref_second ? pr.aux : aux0;
              const int64_t row = tile * 32 + j_;
              const int64_t rc = row < n ? row : 0;
              const float *arow = auxp + (int64_t)(ax.aux_index ? ax.aux_index[rc] : rc) * ax.aux_stride + 4 * h_;
              const int nga = (ax.K0_pad - ax.in_dim) >> 3;
#pragma unroll
              for (int g = 0; g < 8; ++g) avp[g] = *reinterpret_cast<const f32x4 *>(arow + 8 * (g < nga ? g : nga - 1));
          }
      }
      TN_PTG_BEGIN
      f32x4 fr[KP ? 12 : 1];                 // KP: the tile's feature rows as first-layer operands (slots 8g + 4h .. +3)
      if constexpr (KP) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        const float *cr = kp.coords + (valid ? row : 0) * kp.coord_stride;
        const float xs[3] = {cr[0], cr[1], cr[2]};
#pragma unroll
        for (int sc = 0; sc < 3; ++sc) {
            tn::f32x4k prod[4];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float u, v;
                tn::pair_uv(xs, p, u, v);
                const tn::PlaneTaps t = tn::plane_taps(u, v, kp.H[sc], kp.W[sc], 32);
                tn::f32x4k val[4];
                tn::plane_gather<4, 8>(kp.planes[sc][p], t, 4 * h, val);
#pragma unroll
                for (int q = 0; q < 4; ++q) prod[q] = p == 0 ? val[q] : prod[q] * val[q];      // (1*p0)*p1*p2, models.py:157-160
                __builtin_amdgcn_sched_barrier(0);              // one plane's 16 loads in flight at a time (register budget)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fr[4 * sc + q] = f32x4{prod[q][0], prod[q][1], prod[q][2], prod[q][3]};
                if (valid && (STASH || !PAIR || kp.write_feat)) *reinterpret_cast<tn::f32x4k *>(kp.feat + row * 96 + 32 * sc + 8 * q + 4 * h) = prod[q];
            }
        }
      }
      // F2 + KP: the gathered features as fp16 B operands, converted once for both heads (k block b = groups 2 b, 2 b + 1)
      u32x4h xbh[(F2 && KP) ? 6 : 1], xbl[(F2 && KP) ? 6 : 1];
      float inv_x = 1.0f;
      if constexpr (F2 && KP) {
          float m = 0.0f;
#pragma unroll
          for (int g = 0; g < 12; ++g)
#pragma unroll
              for (int u = 0; u < 4; ++u) m = fmaxf(m, fabsf(fr[g][u]));
          float s_x;
          f2_scales(f2_xmax(m), s_x, inv_x);
#pragma unroll
          for (int b = 0; b < 6; ++b) {
              const float v[8] = {fr[2 * b][0], fr[2 * b][1], fr[2 * b][2], fr[2 * b][3], fr[2 * b + 1][0], fr[2 * b + 1][1], fr[2 * b + 1][2], fr[2 * b + 1][3]};
              f2_split8(v, s_x, xbh[b], xbl[b]);
          }
      }
      TN_PTG(8)
      auto head2 = [&](const MlpArgs &a, const float *ldsw, const float *__restrict__ aux, float *__restrict__ y,
                       float *__restrict__ pre_act, float *__restrict__ stash, bool prefetched) {
        if constexpr (F2) {
        const int L = a.n_layers;
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        if constexpr (!STASH) {
            if (a.row_gate != nullptr) {
                const float gate = valid ? a.row_gate[row] : 0.0f;
                if (!__any(gate != 0.0f)) {
                    if (valid && h == 0) {
                        for (int o = 0; o < a.out_dim; ++o) {
                            y[row * a.out_dim + o] = 0.0f;
                            if (pre_act) pre_act[row * a.out_dim + o] = 0.0f;
                        }
                    }
                    return;
                }
            }
        }
        float *stH = nullptr, *stQ = nullptr;
        unsigned *stM = nullptr;
        if constexpr (STASH) {
            const int NH = L - 1;
            stH = stash + tile * (int64_t)(stash_rows(H, NH, 0) * 32);
            stQ = stH + stash_rows_w(H, NH, 0) * 32;
            stM = reinterpret_cast<unsigned *>(stQ + 4 * 32);
        }
        const float *scl = ldsw + a.f2_scale;
        TN_PT_BEGIN
        // ---- layer 0: x columns, then the per-ray table columns, each with its own per-sample scale ----
        const _Float16 *W0 = reinterpret_cast<const _Float16 *>(ldsw + a.w_off[0]);
        const int plane0 = a.f2_plane[0], st0 = a.stride[0];
        const int nbx = a.in_dim >> 4;
        f32x16 acc[T];
#pragma unroll
        for (int ob = 0; ob < T; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][r] = 0.0f;
        float inv0;
        if constexpr (KP) {
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                f2_block(W0, plane0, st0, j, h, b, xbh[b], xbl[b], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            inv0 = inv_x;
        } else {
            // (all loads unconditional from clamped rows, see the fp32 path)
            const float *xr = x + (valid ? row : 0) * a.in_dim + 4 * h;
            float s_x;
            if constexpr (XB > 0) {
                f32x4 xv[2 * XB];
                if (a.x_rows != nullptr) {                   // (wave-uniform) row view of the producer's workspace: feature f of the
                    // tile's sample j at [32 f + j] -- every load one 128-byte row per half-wave (samples >= n: zeros)
                    const float *xt = a.x_rows + tile * a.x_rows_stride + (4 * h) * 32 + j;
#pragma unroll
                    for (int b = 0; b < XB; ++b)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            xv[2 * b][u] = xt[(16 * b + u) * 32];
                            xv[2 * b + 1][u] = xt[(16 * b + 8 + u) * 32];
                        }
                } else {
#pragma unroll
                for (int b = 0; b < XB; ++b) {
                    xv[2 * b] = *reinterpret_cast<const f32x4 *>(xr + 16 * b);
                    xv[2 * b + 1] = *reinterpret_cast<const f32x4 *>(xr + 16 * b + 8);
                }
                }
                float m = 0.0f;
#pragma unroll
                for (int e = 0; e < 2 * XB; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(xv[e][0]), fabsf(xv[e][1]))), fmaxf(fabsf(xv[e][2]), fabsf(xv[e][3])));
                f2_scales(f2_xmax(m), s_x, inv0);
#pragma unroll
                for (int b = 0; b < XB; ++b) {
                    const float v[8] = {xv[2 * b][0], xv[2 * b][1], xv[2 * b][2], xv[2 * b][3], xv[2 * b + 1][0], xv[2 * b + 1][1], xv[2 * b + 1][2], xv[2 * b + 1][3]};
                    u32x4h bh, bl;
                    f2_split8(v, s_x, bh, bl);
                    f2_block(W0, plane0, st0, j, h, b, bh, bl, acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            // pass 1: the column maximum; pass 2: convert + multiply
            float m = 0.0f;
            for (int b = 0; b < nbx; ++b) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(xr + 16 * b), v1 = *reinterpret_cast<const f32x4 *>(xr + 16 * b + 8);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[0]), fabsf(v0[1]))), fmaxf(fmaxf(fabsf(v0[2]), fabsf(v0[3])), fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3])))));
            }
            f2_scales(f2_xmax(m), s_x, inv0);
            f32x4 c0 = *reinterpret_cast<const f32x4 *>(xr), c1 = *reinterpret_cast<const f32x4 *>(xr + 8);
            for (int b = 0; b < nbx; ++b) {
                const int bn = b + 1 < nbx ? b + 1 : b;
                const f32x4 n0 = *reinterpret_cast<const f32x4 *>(xr + 16 * bn), n1 = *reinterpret_cast<const f32x4 *>(xr + 16 * bn + 8);
                const float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                u32x4h bh, bl;
                f2_split8(v, s_x, bh, bl);
                f2_block(W0, plane0, st0, j, h, b, bh, bl, acc);
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1;
            }
            }
        }
        TN_PT(0)
        f32x16 act[T];
        {
            const float c = inv0 * scl[1];
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                tn::pin16(acc[ob]);
#pragma unroll
                for (int r = 0; r < 16; ++r) act[ob][r] = acc[ob][r] * c;
            }
        }
        TN_PT(1)
        if (a.enc == TN_ENC_AUX_CAT) {                       // (wave-uniform)
            const float *arow = aux + (int64_t)(a.aux_index ? a.aux_index[valid ? row : 0] : (valid ? row : 0)) * a.aux_stride + 4 * h;
            const int nga = (a.K0_pad - a.in_dim) >> 3;      // groups of 8 table columns (7 for [PE_8(d), d, 0...])
            f32x4 av[8];
            float m = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {                    // (groups past the table repeat the last one: their weights are zero)
                if (PREF && prefetched) av[g] = avp[PREF ? g : 0];
                else av[g] = *reinterpret_cast<const f32x4 *>(arow + 8 * (g < nga ? g : nga - 1));
                m = fmaxf(fmaxf(m, fmaxf(fabsf(av[g][0]), fabsf(av[g][1]))), fmaxf(fabsf(av[g][2]), fabsf(av[g][3])));
            }
            float s_a, inv_a;
            f2_scales(f2_xmax(m), s_a, inv_a);
#pragma unroll
            for (int ob = 0; ob < T; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ob][r] = 0.0f;
            const int nba = (a.K0_pad - a.in_dim + 15) >> 4;
#pragma unroll
            for (int ba = 0; ba < 4; ++ba) {
                if (ba < nba) {
                    const float v[8] = {av[2 * ba][0], av[2 * ba][1], av[2 * ba][2], av[2 * ba][3], av[2 * ba + 1][0], av[2 * ba + 1][1], av[2 * ba + 1][2], av[2 * ba + 1][3]};
                    u32x4h bh, bl;
                    f2_split8(v, s_a, bh, bl);
                    f2_block(W0, plane0, st0, j, h, nbx + ba, bh, bl, acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const float c = inv_a * scl[1];
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                tn::pin16(acc[ob]);
#pragma unroll
                for (int r = 0; r < 16; ++r) act[ob][r] = fmaf(acc[ob][r], c, act[ob][r]);
            }
        }
#pragma unroll
        for (int ob = 0; ob < T; ++ob) {
            const f32x16 bias = tn::bias_tile(ldsw + a.b_off[0], ob, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) act[ob][r] += bias[r];
            act[ob] = tn::relu16(act[ob]);
            if constexpr (STASH) { stM[ob * 64 + lane] = relu_bits(act[ob]); if constexpr (!LEAN) store_rows(stH, act[ob], ob, j, h); }
        }
        TN_PT(3)
        // ---- hidden layers ----
        for (int l = 1; l + 1 < L; ++l) {
            hidden_layer_f2(ldsw, a, l, act, j, h);
            TN_PT(4)
            if constexpr (STASH) {
#pragma unroll
                for (int ob = 0; ob < T; ++ob) {
                    stM[(l * T + ob) * 64 + lane] = relu_bits(act[ob]);
                    if constexpr (!LEAN) store_rows(stH + l * H * 32, act[ob], ob, j, h);       // (TN_MLP_LEAN leaves the H rows out)
                }
            }
        }
        TN_PT(5)
        // ---- output layer (<= 4 outputs: fp32 dot products on the VALU, as in the fp32 path) ----
        const float *Wf = ldsw + a.w_off[L - 1];
        const float *Bf = ldsw + a.b_off[L - 1];
        const int sf = a.stride[L - 1];
        const int out = a.out_dim;
        float o4[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            o4[o] = 0.f;
            if (o < out) o4[o] = tn::small_out<H>(Wf + o * sf, Bf[o], act, h);
            if constexpr (STASH) { if (h == 0) stQ[o * 32 + j] = o4[o]; }
        }
        if (valid && h == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < out) {
                    if (pre_act) pre_act[row * out + o] = o4[o];
                    y[row * out + o] = tn::apply_act(o4[o], a.out_act);
                }
        }
        TN_PT(6)
        }
      };
      auto head = [&](const MlpArgs &a, const float *ldsw, const float *__restrict__ aux, float *__restrict__ y,
                      float *__restrict__ pre_act, float *__restrict__ stash) {
        const int L = a.n_layers;
        const int G0 = a.K0_pad >> 3;
        // keep per-lane weight addresses out of LICM's reach (hoisted, they cost dozens of VGPRs)
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const int64_t row = tile * 32 + j;
        const bool valid = row < n;
        if constexpr (!STASH) {
            if (a.row_gate != nullptr) {                    // inference: nothing downstream sees rows whose gate is 0
                const float gate = valid ? a.row_gate[row] : 0.0f;
                if (!__any(gate != 0.0f)) {
                    if (valid && h == 0) {
                        for (int o = 0; o < a.out_dim; ++o) {
                            y[row * a.out_dim + o] = 0.0f;
                            if (pre_act) pre_act[row * a.out_dim + o] = 0.0f;
                        }
                    }
                    return;
                }
            }
        }
        const float *xrow = KP ? nullptr : x + (valid ? row : 0) * a.in_dim;
        float aux3[3] = {0.f, 0.f, 0.f};
        const float *auxrow = nullptr;
        if (!FAST && valid) {
            if (a.enc == TN_ENC_POSENC) { aux3[0] = xrow[0]; aux3[1] = xrow[1]; aux3[2] = xrow[2]; }
            else if (a.enc == TN_ENC_DIR_CAT) { aux3[0] = aux[3 * row]; aux3[1] = aux[3 * row + 1]; aux3[2] = aux[3 * row + 2]; }
            else if (a.enc == TN_ENC_AUX_CAT) auxrow = aux + (int64_t)(a.aux_index ? a.aux_index[row] : row) * a.aux_stride;
        }
        float *stH = nullptr, *stE = nullptr, *stQ = nullptr;
        unsigned *stM = nullptr;
        int xs = 0, extra = 0;
        if constexpr (STASH) {
            const int NH = L - 1;
            extra = extra_rows(a.enc, a.in_dim, a.K0_pad);
            xs = x_slots(a.enc, a.in_dim);
            stH = stash + tile * (int64_t)(stash_rows(H, NH, extra) * 32);
            stE = stH + (2 * NH * H + 4) * 32;
            stQ = stH + stash_rows_w(H, NH, extra) * 32;
            stM = reinterpret_cast<unsigned *>(stQ + 4 * 32);
        }
        // ---- layer 0: inputs streamed 4 slots at a time, prefetched one group ahead ----
        const float *W0 = WLDS ? ldsw + a.w_off[0] : a.W[0];
        const float *B0 = WLDS ? ldsw + a.b_off[0] : a.B[0];
        f32x16 act[T];
#pragma unroll
        for (int ob = 0; ob < T; ++ob) act[ob] = tn::bias_tile(B0, ob, h);
        // Plain x columns (+ aux-table columns): the inputs are loaded UNCONDITIONALLY from clamped addresses, two groups
        // ahead.  (A load whose result is merged with a constant -- `valid ? x[..] : 0` -- makes the compiler wait for
        // it at the merge, which turns the prefetch into a blocking load per group.)  Rows past n read row 0, slots past
        // the input width read slot 0: both only ever meet zero weights or discarded outputs.
        if constexpr (FAST) {
            const float *arow = a.enc == TN_ENC_AUX_CAT
                                    ? aux + (int64_t)(a.aux_index ? a.aux_index[valid ? row : 0] : (valid ? row : 0)) * a.aux_stride
                                    : (KP ? kp.feat : xrow);
            auto in_ptr = [&](int g) {
                const int q0 = 8 * (g < G0 ? g : G0 - 1) + 4 * h;
                // (KP: the x slots live in registers; a head without aux columns requests a dummy line it never uses)
                const float *xp = KP ? kp.feat : xrow + q0;
                const float *p = q0 < a.in_dim ? xp : (a.enc == TN_ENC_AUX_CAT ? arow + (q0 - a.in_dim) : (KP ? kp.feat : xrow));
                return reinterpret_cast<const f32x4 *>(p);
            };
            // (KP: the aux-table slots behind the features are requested before the 12 register groups run)
            f32x4 b0 = *in_ptr(KP ? 12 : 0), b1 = *in_ptr(KP ? 13 : 1);
            f32x4 w[T], wn[T];
#pragma unroll
            for (int ob = 0; ob < T; ++ob) w[ob] = load_a4<true>(W0, 32 * ob + j, 4 * h, a.K0, a.stride[0]);
            int g_first = 0;
            if constexpr (KP) {                 // slots 0..95: the gathered features, already in registers
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    const int gn = g + 1 < G0 ? g + 1 : g;
#pragma unroll
                    for (int ob = 0; ob < T; ++ob) wn[ob] = load_a4<true>(W0, 32 * ob + j, 8 * gn + 4 * h, a.K0, a.stride[0]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int ob = 0; ob < T; ++ob) act[ob] = tn::mfma32(w[ob][u], fr[g][u], act[ob]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ob = 0; ob < T; ++ob) w[ob] = wn[ob];
                }
                g_first = 12;
                // the loop below may not run at all (head without aux columns): results are materialised HERE, in straight-line
                // code behind the last MFMA (see tn::pin16 -- hipcc under-counts the MFMA wait states on the short path)
#pragma unroll
                for (int ob = 0; ob < T; ++ob) tn::pin16(act[ob]);
            }
            for (int g = g_first; g < G0; ++g) {
                const f32x4 b2 = *in_ptr(g + 2);
                const int gn = g + 1 < G0 ? g + 1 : g;           // weights of the next group, requested before this group's MFMAs
#pragma unroll
                for (int ob = 0; ob < T; ++ob) wn[ob] = load_a4<true>(W0, 32 * ob + j, 8 * gn + 4 * h, a.K0, a.stride[0]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ob = 0; ob < T; ++ob) act[ob] = tn::mfma32(w[ob][u], b0[u], act[ob]);
                __builtin_amdgcn_sched_barrier(0);
                b0 = b1; b1 = b2;
#pragma unroll
                for (int ob = 0; ob < T; ++ob) w[ob] = wn[ob];
            }
        } else {
        f32x4 b = fetch_input(a, xrow, aux3, valid, 0, h, auxrow);
        for (int g = 0; g < G0; ++g) {
            f32x4 bn = {0.f, 0.f, 0.f, 0.f};
            if (g + 1 < G0) bn = fetch_input(a, xrow, aux3, valid, g + 1, h, auxrow);
            f32x4 w[T];
#pragma unroll
            for (int ob = 0; ob < T; ++ob) {
                if constexpr (WLDS) w[ob] = load_a4<true>(W0, 32 * ob + j, 8 * g + 4 * h, a.K0, a.stride[0]);
                else if (a.enc != TN_ENC_DIR_CAT && a.enc != TN_ENC_AUX_CAT) w[ob] = load_a4<false>(W0, 32 * ob + j, 8 * g + 4 * h, a.K0, a.K0);
                else {   // global first layer of a dir_cat head: slot -> torch column, one dword at a time
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = 8 * g + 4 * h + u;
                        w[ob][u] = q < a.K0 ? W0[(int64_t)(32 * ob + j) * a.K0 + layer0_col(a, q)] : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ob = 0; ob < T; ++ob) act[ob] = tn::mfma32(w[ob][u], b[u], act[ob]);
            if constexpr (STASH) {
                if (extra > 0 && 8 * g + 4 * h + 3 >= xs) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = 8 * g + 4 * h + u;
                        if (q >= xs) stE[(q - xs) * 32 + j] = b[u];
                    }
                }
            }
            b = bn;
        }
        }
#pragma unroll
        for (int ob = 0; ob < T; ++ob) {
            tn::pin16(act[ob]);
            act[ob] = tn::relu16(act[ob]);
            if constexpr (STASH) { stM[ob * 64 + lane] = relu_bits(act[ob]); store_rows(stH, act[ob], ob, j, h); }
        }
        // ---- hidden layers H -> H ----
        for (int l = 1; l + 1 < L; ++l) {
            const float *Wl = WLDS ? ldsw + a.w_off[l] : a.W[l];
            const float *Bl = WLDS ? ldsw + a.b_off[l] : a.B[l];
            tn::hidden_layer<H>(Wl, Bl, WLDS ? a.stride[l] : H, act, j, h);
            if constexpr (STASH) {
#pragma unroll
                for (int ob = 0; ob < T; ++ob) {
                    stM[(l * T + ob) * 64 + lane] = relu_bits(act[ob]);
                    store_rows(stH + l * H * 32, act[ob], ob, j, h);
                }
            }
        }
        // ---- output layer ----
        const float *Wf = WLDS ? ldsw + a.w_off[L - 1] : a.W[L - 1];
        const float *Bf = WLDS ? ldsw + a.b_off[L - 1] : a.B[L - 1];
        const int sf = WLDS ? a.stride[L - 1] : H;
        const int out = a.out_dim;
        if (out <= 4) {
            float o4[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                o4[o] = 0.f;
                if (o < out) o4[o] = tn::small_out<H>(Wf + o * sf, Bf[o], act, h);
                if constexpr (STASH) { if (h == 0) stQ[o * 32 + j] = o4[o]; }
            }
            if (valid && h == 0) {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < out) {
                        if (pre_act) pre_act[row * out + o] = o4[o];
                        y[row * out + o] = tn::apply_act(o4[o], a.out_act);
                    }
            }
        } else {
            const int n_ob = (out + 31) >> 5;
            for (int ob = 0; ob < n_ob; ++ob) {
                f32x16 acc;
                if constexpr (WLDS) acc = tn::bias_tile(Bf, ob, h);
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
                        acc[r] = f < out ? Bf[f] : 0.f;
                    }
                }
                const int arow = 32 * ob + j;
#pragma unroll
                for (int kb = 0; kb < T; ++kb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 w = {0.f, 0.f, 0.f, 0.f};
                        if (WLDS || arow < out) w = load_a4<WLDS>(Wf, arow, 32 * kb + 8 * q + 4 * h, H, sf);
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc = tn::mfma32(w[u], act[kb][4 * q + u], acc);
                    }
                }
                tn::pin16(acc);
                if (valid) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * ob + 8 * q + 4 * h;
                        if (f0 + 3 < out && (out & 3) == 0) {
                            f32x4 v;
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = acc[4 * q + u];
                            if (pre_act) *reinterpret_cast<f32x4 *>(pre_act + row * out + f0) = v;
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = tn::apply_act(v[u], a.out_act);
                            *reinterpret_cast<f32x4 *>(y + row * out + f0) = v;
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (f0 + u < out) {
                                    if (pre_act) pre_act[row * out + f0 + u] = acc[4 * q + u];
                                    y[row * out + f0 + u] = tn::apply_act(acc[4 * q + u], a.out_act);
                                }
                        }
                    }
                }
            }
        }
      };
      if constexpr (F2) {
          head2(a0, lds, aux0, y0, pre_act0, stash0, !pref_second);
          if constexpr (PAIR) head2(pr.b, lds + a0.lds_floats, pr.aux, pr.y, nullptr, pr.stash, pref_second);
          TN_PTG(9)
      } else {
          head(a0, lds, aux0, y0, pre_act0, stash0);
          if constexpr (PAIR) head(pr.b, lds + a0.lds_floats, pr.aux, pr.y, nullptr, pr.stash);
      }
    }
}

// reference models.py:30-39 as a stand-alone kernel (PositionalEncoding.forward)
__global__ void posenc_kernel(const float *__restrict__ x, int64_t n, int C, const float *__restrict__ freqs, int F,
                              float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int width = 2 * F * C;
    if (e >= n * width) return;
    const int64_t row = e / width;
    const int p = (int)(e - row * width);
    const int c = p / (2 * F);
    const int rem = p - c * 2 * F;
    const bool is_cos = rem >= F;
    const int f = is_cos ? rem - F : rem;
    const float ang = x[row * C + c] * (freqs ? freqs[f] : ldexpf(3.14159274101257324f, f));
    out[e] = is_cos ? cosf(ang) : sinf(ang);
}

template <int H>
int launch_fwd(const MlpArgs &a_in, const float *x, const float *aux, int64_t n, float *y, float *pre_act, float *stash, hipStream_t s,
               const FwdPair *pair = nullptr, const KpFwd *kp = nullptr)
{
    const int64_t n_tiles = (n + 31) / 32;
    MlpArgs a = a_in;
    FwdPair pr;
    pr.aux = nullptr; pr.y = nullptr; pr.stash = nullptr;
    if (pair) pr = *pair;
    // f16x2 heads (TN_MLP_F16X2, mlp_f2_heads.h): same launch, the weights staged as fp16 hi / lo planes
    bool f2 = false;
    if constexpr (H == 64) {
        f2 = f2_head_ok(a, H) && (!pair || f2_head_ok(pr.b, H)) && (!kp || a.in_dim == 96);
        if (f2) { plan_f2(a, H); if (pair) plan_f2(pr.b, H); }
    }
    size_t lds_bytes = (size_t)a.lds_floats * 4;
    if (pair) lds_bytes += (size_t)pr.b.lds_floats * 4; else pr.b = a;
    if (f2 && lds_bytes > (size_t)LDS_LIMIT_BYTES) {       // (cannot happen for the reference's heads; keep the fp32 form otherwise)
        f2 = false; a = a_in; if (pair) pr = *pair; else pr.b = a;
        lds_bytes = (size_t)a.lds_floats * 4 + (pair ? (size_t)pr.b.lds_floats * 4 : 0);
    }
    const bool wlds = lds_bytes <= (size_t)LDS_LIMIT_BYTES && a.enc != -1;
    auto plain_cols = [](const MlpArgs &m) { return m.enc == TN_ENC_AUX_CAT || (m.enc == TN_ENC_NONE && (m.in_dim & 3) == 0); };
    const bool fast = wlds && plain_cols(a) && (!pair || plain_cols(pr.b));
    constexpr int WPB = H <= 64 ? 16 : 4;     // 16 waves share one LDS copy of the weights: 4 waves per SIMD
    if (a.x_from_rows && !(wlds && stash && H == 64 && !kp))
        return tn::fail(TN_E_CONFIG, "mlp forward: TN_MLP_X_FROM_ROWS is only read by tn_mlp_fwd_stash of a width-64 f16x2 head");
    if (wlds && stash) {
        if constexpr (H == 64) {
            // stash variants: 12 waves (170-VGPR budget) for the generic first layer, 16 for the plain-column one (119 VGPRs)
            auto launch = [&](auto kern, int wps) -> int {
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e != hipSuccess) { tn::set_error("mlp: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(LDS_LIMIT_BYTES / lds_bytes, 2048 / (wps * 64)));
                const int64_t blocks = std::min<int64_t>((n_tiles + wps - 1) / wps, 256 * per_cu);
                kern<<<dim3((unsigned)blocks), dim3(wps * 64), lds_bytes, s>>>(a, x, aux, n, y, pre_act, stash, pr, KpFwd());
                return TN_OK;
            };
            int rc;
            if (kp) {                   // gather fused in: 12 waves (170-VGPR budget: 48 registers hold the tile's features)
                if (!(fast && pair)) return tn::fail(TN_E_CONFIG, "tn_kplanes_mlp_fwd_pair: both heads must take the plain-column first layer");
                auto go = [&](auto kern, int wv) -> int {
                    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                    if (e != hipSuccess) { tn::set_error("mlp: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
                    const int64_t blocks = std::min<int64_t>((n_tiles + wv - 1) / wv, 256);
                    kern<<<dim3((unsigned)blocks), dim3(wv * 64), lds_bytes, s>>>(a, x, aux, n, y, pre_act, stash, pr, *kp);
                    return TN_OK;
                };
                // fp32 heads: 12 waves (8 waves x 256 VGPRs measured the same: 0.86 ms)
                if (a.lean && !f2) return tn::fail(TN_E_CONFIG, "TN_MLP_LEAN needs the f16x2 heads (TN_MLP_F16X2)");
                const int rc = f2 ? (a.lean ? go(mlp_fwd_kernel<H, true, TN_F2_KP_WAVES, true, true, true, true, true, 0, true>, TN_F2_KP_WAVES)
                                            : go(mlp_fwd_kernel<H, true, TN_F2_KP_WAVES, true, true, true, true, true>, TN_F2_KP_WAVES))
                                  : go(mlp_fwd_kernel<H, true, 12, true, true, true, true>, 12);
                if (rc) return rc;
                return tn::check_launch("mlp_fwd_kernel(kplanes)");
            }
            if (a.x_from_rows && !(f2 && !pair && a.x_rows != nullptr && (a.in_dim == 256 || a.in_dim == 128)))
                return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_stash: TN_MLP_X_FROM_ROWS needs an f16x2 head with 128 or 256 inputs and x_rows");
            if (a.lean && !(f2 && pair)) return tn::fail(TN_E_CONFIG, "TN_MLP_LEAN needs the paired f16x2 heads (TN_MLP_F16X2, tn_mlp_fwd_stash_pair)");
            if (a.lean) rc = launch(mlp_fwd_kernel<H, true, 12, true, true, true, false, true, 0, true>, 12);
            else if (f2 && !pair && a.in_dim == 256) rc = launch(mlp_fwd_kernel<H, true, 8, true, false, true, false, true, 16>, 8);      // (x: 128 VGPRs)
            else if (f2 && !pair && a.in_dim == 128) rc = launch(mlp_fwd_kernel<H, true, 12, true, false, true, false, true, 8>, 12);
            else if (f2) rc = pair ? launch(mlp_fwd_kernel<H, true, 12, true, true, true, false, true>, 12) : launch(mlp_fwd_kernel<H, true, 12, true, false, true, false, true>, 12);
            else if (fast) rc = pair ? launch(mlp_fwd_kernel<H, true, 16, true, true, true>, 16) : launch(mlp_fwd_kernel<H, true, 16, true, false, true>, 16);
            else rc = pair ? launch(mlp_fwd_kernel<H, true, 12, true, true, false>, 12) : launch(mlp_fwd_kernel<H, true, 12, true, false, false>, 12);
            if (rc) return rc;
        } else return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_stash: the register-resident training forward is built for width 64");
    } else if (wlds && kp) {      // inference: gather + one head (the sigma head; the colour head then reads the feature rows where w > 0)
        if constexpr (H == 64) {          // ... or gather + BOTH heads, nothing stashed, no feature rows (pair != nullptr)
            if (!fast) return tn::fail(TN_E_CONFIG, "tn_kplanes_mlp_fwd: the head must take the plain-column first layer");
            auto go = [&](auto kern, int wv) -> int {
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e != hipSuccess) { tn::set_error("mlp: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(LDS_LIMIT_BYTES / lds_bytes, 2048 / (wv * 64)));
                const int64_t blocks = std::min<int64_t>((n_tiles + wv - 1) / wv, 256 * per_cu);
                kern<<<dim3((unsigned)blocks), dim3(wv * 64), lds_bytes, s>>>(a, x, aux, n, y, pre_act, stash, pr, *kp);
                return TN_OK;
            };
            // (the f16x2 pair holds the fp16 features beside both heads' tiles: 8 waves x 256 VGPRs, as in training)
            const int rc = f2 ? (pair ? go(mlp_fwd_kernel<H, true, TN_F2_KP_WAVES, false, true, true, true, true>, TN_F2_KP_WAVES)
                                      : go(mlp_fwd_kernel<H, true, 12, false, false, true, true, true>, 12))
                              : (pair ? go(mlp_fwd_kernel<H, true, 12, false, true, true, true>, 12)
                                      : go(mlp_fwd_kernel<H, true, 12, false, false, true, true>, 12));
            if (rc) return rc;
            return tn::check_launch("mlp_fwd_kernel(kplanes, inference)");
        } else return tn::fail(TN_E_CONFIG, "tn_kplanes_mlp_fwd: width-64 heads only");
    } else if (wlds) {
        auto kern = fast ? mlp_fwd_kernel<H, true, WPB, false, false, true> : mlp_fwd_kernel<H, true, WPB, false, false, false>;
        if constexpr (H == 64) { if (f2) kern = mlp_fwd_kernel<H, true, WPB, false, false, true, false, true>; }
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { tn::set_error("mlp: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(LDS_LIMIT_BYTES / lds_bytes, 2048 / (WPB * 64)));
        const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * per_cu);
        kern<<<dim3((unsigned)blocks), dim3(WPB * 64), lds_bytes, s>>>(a, x, aux, n, y, pre_act, stash, pr, KpFwd());
    } else {
        if (stash) return tn::fail(TN_E_CONFIG, "tn_mlp_fwd_stash: weights must fit LDS");
        // weights streamed from L2
        auto kern = mlp_fwd_kernel<H, false, WPB, false>;
        const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256 * 2);
        kern<<<dim3((unsigned)blocks), dim3(WPB * 64), 0, s>>>(a, x, aux, n, y, pre_act, nullptr, pr, KpFwd());
    }
    return tn::check_launch("mlp_fwd_kernel");
}

// aux table of TN_ENC_AUX_CAT for the colour head: [PE(d), d, 0...] per ray
__global__ void dir_encode_kernel(const float *__restrict__ d, int64_t n, const float *__restrict__ freqs, int F, float *__restrict__ out, int stride)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * stride) return;
    const int64_t row = e / stride;
    const int p = (int)(e - row * stride);
    const float dc[3] = {d[3 * row], d[3 * row + 1], d[3 * row + 2]};
    float v = 0.0f;
    if (p < 6 * F) v = tn::posenc_value(dc, p, F, freqs);
    else if (p < 6 * F + 3) v = dc[p - 6 * F];
    out[e] = v;
}

int fwd_common(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, float *pre_act, float *stash,
               hipStream_t s, const char *who, const FwdPair *pair = nullptr, const KpFwd *kp = nullptr)
{
    MlpArgs a;
    int H = 0;
    if (int rc = plan(desc, a, H)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_mlp_fwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE((x || kp) && y, TN_E_NULL, "tn_mlp_fwd: null pointer");
    TN_REQUIRE((a.enc != TN_ENC_DIR_CAT && a.enc != TN_ENC_AUX_CAT) || aux, TN_E_NULL, "tn_mlp_fwd: dir_cat / aux_cat need aux");
    TN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, TN_E_ALIGN, "tn_mlp_fwd: x / y must be 16-byte aligned");
    TN_REQUIRE(!kp || H == 64, TN_E_CONFIG, "tn_kplanes_mlp_fwd_pair: width-64 heads only");
    TN_REQUIRE(a.enc != TN_ENC_AUX_CAT || ((uintptr_t)aux & 15) == 0, TN_E_ALIGN, "tn_mlp_fwd: aux table must be 16-byte aligned");
    switch (H) {
    case 32: return launch_fwd<32>(a, x, aux, n, y, pre_act, stash, s, pair);
    case 64: return launch_fwd<64>(a, x, aux, n, y, pre_act, stash, s, pair, kp);
    case 128: return launch_fwd<128>(a, x, aux, n, y, pre_act, stash, s, pair);
    default: return launch_fwd<256>(a, x, aux, n, y, pre_act, stash, s, pair);
    }
}

}  // namespace

extern "C" int tn_mlp_fwd(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, float *pre_act,
                          void *stream)
{
    return fwd_common(desc, x, aux, n, y, pre_act, nullptr, (hipStream_t)stream, "tn_mlp_fwd");
}

extern "C" int64_t tn_mlp_bwd_workspace_bytes(const tn_mlp_desc *desc, int64_t n);

extern "C" int tn_mlp_fwd_stash_layers(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, float *workspace,
                                       void *stream);

extern "C" int tn_mlp_fwd_stash(const tn_mlp_desc *desc, const float *x, const float *aux, int64_t n, float *y, void *workspace,
                                int64_t workspace_bytes, void *stream)
{
    TN_REQUIRE(desc, TN_E_NULL, "tn_mlp_fwd_stash: null descriptor");
    const int H = desc->dims[1];
    if (n <= 0) return n == 0 ? TN_OK : tn::fail(TN_E_SIZE, "tn_mlp_fwd_stash: negative n");
    if (!two_pass_supported(desc)) {       // wide / deep stacks: the layer-by-layer backward's workspace
        const int64_t need_l = tn_mlp_bwd_workspace_bytes(desc, n);
        TN_REQUIRE(need_l > 0, TN_E_CONFIG, "tn_mlp_fwd_stash: this configuration's backward does not use a workspace");
        TN_REQUIRE(workspace && workspace_bytes >= need_l, TN_E_NULL, "tn_mlp_fwd_stash: workspace missing or too small");
        TN_REQUIRE(((uintptr_t)workspace & 15) == 0, TN_E_ALIGN, "tn_mlp_fwd_stash: workspace must be 16-byte aligned");
        return tn_mlp_fwd_stash_layers(desc, x, aux, n, y, (float *)workspace, stream);
    }
    const int64_t need = tn_mlp_bwd_workspace_bytes(desc, n);
    const int extra = extra_rows(desc->encoding, desc->in_dim, (desc->dims[0] + 7) & ~7);
    TN_REQUIRE(need == ((n + 31) / 32) * (int64_t)stash_rows(H, desc->n_layers - 1, extra) * 128, TN_E_CONFIG,
               "tn_mlp_fwd_stash: configuration not covered by the two-pass backward");
    TN_REQUIRE(workspace && workspace_bytes >= need, TN_E_NULL, "tn_mlp_fwd_stash: workspace missing or too small");
    TN_REQUIRE(((uintptr_t)workspace & 15) == 0, TN_E_ALIGN, "tn_mlp_fwd_stash: workspace must be 16-byte aligned");
    return fwd_common(desc, x, aux, n, y, nullptr, (float *)workspace, (hipStream_t)stream, "tn_mlp_fwd_stash");
}

extern "C" int tn_mlp_fwd_stash_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux, int64_t n,
                                     float *y, float *partner_y, void *workspace, int64_t workspace_bytes, void *partner_workspace,
                                     int64_t partner_workspace_bytes, void *stream)
{
    TN_REQUIRE(desc && partner, TN_E_NULL, "tn_mlp_fwd_stash_pair: null descriptor");
    const int H = desc->dims[1];
    TN_REQUIRE(H == 64 && partner->dims[1] == 64 && desc->n_layers >= 2 && desc->n_layers <= 5 && partner->n_layers >= 2 &&
                   partner->n_layers <= 5 && desc->dims[desc->n_layers] <= 4 && partner->dims[partner->n_layers] <= 4 &&
                   partner->in_dim == desc->in_dim && (desc->in_dim & 3) == 0 &&
                   (desc->encoding == TN_ENC_NONE || desc->encoding == TN_ENC_AUX_CAT) && partner->encoding == TN_ENC_NONE,
               TN_E_CONFIG, "tn_mlp_fwd_stash_pair: two width-64 heads (<= 5 layers, <= 4 outputs) on the same x; partner without encoding");
    if (n <= 0) return n == 0 ? TN_OK : tn::fail(TN_E_SIZE, "tn_mlp_fwd_stash_pair: negative n");
    const int64_t need_a = tn_mlp_bwd_workspace_bytes(desc, n), need_b = tn_mlp_bwd_workspace_bytes(partner, n);
    TN_REQUIRE(need_a > 0 && need_b > 0, TN_E_CONFIG, "tn_mlp_fwd_stash_pair: configuration not covered by the two-pass backward");
    TN_REQUIRE(workspace && workspace_bytes >= need_a && partner_workspace && partner_workspace_bytes >= need_b, TN_E_NULL,
               "tn_mlp_fwd_stash_pair: workspace missing or too small");
    TN_REQUIRE((((uintptr_t)workspace | (uintptr_t)partner_workspace | (uintptr_t)partner_y) & 15) == 0, TN_E_ALIGN,
               "tn_mlp_fwd_stash_pair: buffers must be 16-byte aligned");
    TN_REQUIRE(partner_y, TN_E_NULL, "tn_mlp_fwd_stash_pair: null pointer");
    FwdPair pr;
    int Hb = 0;
    if (int rc = plan(partner, pr.b, Hb)) return rc;
    pr.aux = nullptr; pr.y = partner_y; pr.stash = (float *)partner_workspace;
    return fwd_common(desc, x, aux, n, y, nullptr, (float *)workspace, (hipStream_t)stream, "tn_mlp_fwd_stash_pair", &pr);
}

static int kp_fwd_args(const tn_kplanes_desc *kd, const float *coords, int64_t coord_stride, float *feat, KpFwd &kp, const char *who)
{
    TN_REQUIRE(kd, TN_E_NULL, "tn_kplanes_mlp_fwd: null descriptor");
    TN_REQUIRE(kd->n_scales == 3 && kd->channels == 32, TN_E_CONFIG, "tn_kplanes_mlp_fwd: 3 scales x 32 channels (run.py:136)");
    TN_REQUIRE(coord_stride >= 3, TN_E_SIZE, "tn_kplanes_mlp_fwd: bad coordinate stride");
    TN_REQUIRE(coords && feat, TN_E_NULL, "tn_kplanes_mlp_fwd: null pointer");
    TN_REQUIRE(((uintptr_t)feat & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd: feat must be 16-byte aligned");
    for (int s = 0; s < 3; ++s) {
        TN_REQUIRE(kd->height[s] > 0 && kd->width[s] > 0 && (int64_t)kd->height[s] * kd->width[s] * 32 < (1ll << 30), TN_E_SIZE,
                   "tn_kplanes_mlp_fwd: bad plane resolution");
        kp.H[s] = kd->height[s]; kp.W[s] = kd->width[s];
        for (int p = 0; p < 3; ++p) {
            TN_REQUIRE(kd->planes[s][p], TN_E_NULL, "tn_kplanes_mlp_fwd: null plane pointer");
            TN_REQUIRE(((uintptr_t)kd->planes[s][p] & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd: planes must be 16-byte aligned");
            kp.planes[s][p] = kd->planes[s][p];
        }
    }
    kp.coords = coords; kp.coord_stride = coord_stride; kp.feat = feat;
    return TN_OK;
}

extern "C" int tn_kplanes_mlp_fwd(const tn_kplanes_desc *kd, const float *coords, int64_t coord_stride, const tn_mlp_desc *desc, int64_t n,
                                  float *feat, float *y, void *stream)
{
    TN_REQUIRE(desc, TN_E_NULL, "tn_kplanes_mlp_fwd: null descriptor");
    TN_REQUIRE(desc->in_dim == 96 && desc->encoding == TN_ENC_NONE && desc->dims[1] == 64 && desc->row_gate == nullptr, TN_E_CONFIG,
               "tn_kplanes_mlp_fwd: an ungated width-64 head without encoding on the 96 features");
    if (n <= 0) return n == 0 ? TN_OK : tn::fail(TN_E_SIZE, "tn_kplanes_mlp_fwd: negative n");
    KpFwd kp;
    if (int rc = kp_fwd_args(kd, coords, coord_stride, feat, kp, "tn_kplanes_mlp_fwd")) return rc;
    TN_REQUIRE(y && ((uintptr_t)y & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd: y must be a 16-byte aligned pointer");
    return fwd_common(desc, nullptr, nullptr, n, y, nullptr, nullptr, (hipStream_t)stream, "tn_kplanes_mlp_fwd", nullptr, &kp);
}

extern "C" int tn_kplanes_mlp_fwd_pair(const tn_kplanes_desc *kd, const float *coords, int64_t coord_stride, const tn_mlp_desc *desc,
                                       const tn_mlp_desc *partner, const float *aux, int64_t n, float *feat, float *y, float *partner_y,
                                       void *workspace, int64_t workspace_bytes, void *partner_workspace, int64_t partner_workspace_bytes,
                                       void *stream)
{
    TN_REQUIRE(kd && desc && partner, TN_E_NULL, "tn_kplanes_mlp_fwd_pair: null descriptor");
    TN_REQUIRE(kd->n_scales == 3 && kd->channels == 32 && desc->in_dim == 96 && partner->in_dim == 96, TN_E_CONFIG,
               "tn_kplanes_mlp_fwd_pair: 3 scales x 32 channels feeding two heads with in_dim 96 (run.py:136-139)");
    TN_REQUIRE(coord_stride >= 3, TN_E_SIZE, "tn_kplanes_mlp_fwd_pair: bad coordinate stride");
    if (n <= 0) return n == 0 ? TN_OK : tn::fail(TN_E_SIZE, "tn_kplanes_mlp_fwd_pair: negative n");
    const bool inference = workspace == nullptr && partner_workspace == nullptr;
    TN_REQUIRE(coords && (feat || inference), TN_E_NULL, "tn_kplanes_mlp_fwd_pair: null pointer");
    TN_REQUIRE(((uintptr_t)feat & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd_pair: feat must be 16-byte aligned");
    TN_REQUIRE(!inference || ((uintptr_t)coords & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd_pair: coords must be 16-byte aligned");
    KpFwd kp;
    for (int s = 0; s < 3; ++s) {
        TN_REQUIRE(kd->height[s] > 0 && kd->width[s] > 0 && (int64_t)kd->height[s] * kd->width[s] * 32 < (1ll << 30), TN_E_SIZE,
                   "tn_kplanes_mlp_fwd_pair: bad plane resolution");
        kp.H[s] = kd->height[s]; kp.W[s] = kd->width[s];
        for (int p = 0; p < 3; ++p) {
            TN_REQUIRE(kd->planes[s][p], TN_E_NULL, "tn_kplanes_mlp_fwd_pair: null plane pointer");
            TN_REQUIRE(((uintptr_t)kd->planes[s][p] & 15) == 0, TN_E_ALIGN, "tn_kplanes_mlp_fwd_pair: planes must be 16-byte aligned");
            kp.planes[s][p] = kd->planes[s][p];
        }
    }
    kp.coords = coords; kp.coord_stride = coord_stride; kp.feat = feat;
    if (inference) {        // no backward follows: no workspace rows, no feature rows (the dummy operand line points at the coordinates)
        kp.write_feat = 0;
        if (!feat) {
            TN_REQUIRE(n * coord_stride >= 4, TN_E_SIZE, "tn_kplanes_mlp_fwd_pair: inference form without feat needs >= 16 B of coordinates");
            kp.feat = const_cast<float *>(coords);
        }
    }
    // the rest is tn_mlp_fwd_stash_pair with the x rows replaced by the gather
    const int H = desc->dims[1];
    TN_REQUIRE(H == 64 && partner->dims[1] == 64 && desc->n_layers >= 2 && desc->n_layers <= 5 && partner->n_layers >= 2 &&
                   partner->n_layers <= 5 && desc->dims[desc->n_layers] <= 4 && partner->dims[partner->n_layers] <= 4 &&
                   (desc->encoding == TN_ENC_NONE || desc->encoding == TN_ENC_AUX_CAT) && partner->encoding == TN_ENC_NONE,
               TN_E_CONFIG, "tn_kplanes_mlp_fwd_pair: two width-64 heads (<= 5 layers, <= 4 outputs); partner without encoding");
    const int64_t need_a = tn_mlp_bwd_workspace_bytes(desc, n), need_b = tn_mlp_bwd_workspace_bytes(partner, n);
    TN_REQUIRE(need_a > 0 && need_b > 0, TN_E_CONFIG, "tn_kplanes_mlp_fwd_pair: configuration not covered by the two-pass backward");
    TN_REQUIRE(y && partner_y && (inference || (workspace && workspace_bytes >= need_a && partner_workspace && partner_workspace_bytes >= need_b)),
               TN_E_NULL, "tn_kplanes_mlp_fwd_pair: workspace / output missing or too small (both workspaces NULL: inference form)");
    TN_REQUIRE(!inference || (desc->row_gate == nullptr && partner->row_gate == nullptr), TN_E_CONFIG, "tn_kplanes_mlp_fwd_pair: no row gates in the pair form");
    TN_REQUIRE((((uintptr_t)workspace | (uintptr_t)partner_workspace | (uintptr_t)partner_y | (uintptr_t)y) & 15) == 0, TN_E_ALIGN,
               "tn_kplanes_mlp_fwd_pair: buffers must be 16-byte aligned");
    FwdPair pr;
    int Hb = 0;
    if (int rc = plan(partner, pr.b, Hb)) return rc;
    pr.aux = nullptr; pr.y = partner_y; pr.stash = (float *)partner_workspace;
    return fwd_common(desc, nullptr, aux, n, y, nullptr, (float *)workspace, (hipStream_t)stream, "tn_kplanes_mlp_fwd_pair", &pr, &kp);
}

extern "C" int tn_dir_encode(const float *dirs, int64_t n, const float *freqs, int n_freqs, float *out, int stride, void *stream)
{
    TN_REQUIRE(n >= 0 && n_freqs > 0 && stride >= 6 * n_freqs + 3, TN_E_SIZE, "tn_dir_encode: bad size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(dirs && out, TN_E_NULL, "tn_dir_encode: null pointer");
    const int64_t total = n * stride;
    dir_encode_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(dirs, n, freqs, n_freqs, out, stride);
    return tn::check_launch("dir_encode_kernel");
}

extern "C" int tn_posenc_fwd(const float *x, int64_t n, int n_channels, const float *freqs, int n_freqs, float *out,
                             void *stream)
{
    TN_REQUIRE(n >= 0 && n_channels > 0 && n_freqs > 0, TN_E_SIZE, "tn_posenc_fwd: bad size");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && out, TN_E_NULL, "tn_posenc_fwd: null pointer");
    const int64_t total = n * 2 * n_freqs * n_channels;
    posenc_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, n, n_channels, freqs, n_freqs, out);
    return tn::check_launch("posenc_kernel");
}
