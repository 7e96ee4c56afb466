// Weight gradients of the paired width-64 heads WITHOUT a stash of hidden activations (TN_MLP_LEAN, round 5).
//
// Reference: the autograd of the two decoders (src/models.py:7-28 MLP, :70-77 VanillaOpacityDecoder, :79-89 VanillaColorDecoder):
//     dW_l = sum_s G_l[:, s] H_l[:, s]^T,   db_l = sum_s G_l[:, s]          (H_0 = the encoded input, G_l = d loss / d (W_l H_l + b_l))
//
// Rounds 1-4 had the training forward write every H_l as [feature][32-sample] rows (1.3 KB per sample) and the weight-gradient
// kernels read them back: 2.7 GB of the K-Planes step's 11.9 GB, from a trade (recompute = 344 fp32 MFMAs per tile) that the f16x2
// forward has made obsolete.  Here H never crosses HBM:
//
//   wgrad_first_kernel   dW_0 / db_0 of BOTH heads from the feature rows x [n, 96], the per-ray table and the chain's G_0 rows.
//                        A wave owns one 32-column block of the first layers' inputs (three x blocks, two table blocks) and the
//                        32 x 32 tiles of both heads on it; both MFMA operands are read from memory in operand layout (G rows:
//                        lane = feature, 32 contiguous bytes per k block; x: lane = column, one coalesced 128-byte row segment per
//                        sample) -- no LDS, no barrier.
//   wgrad_rc_kernel      dW_1 .. dW_4, dW_1 of the sigma head and their biases.  A wave owns 32 samples and REBUILDS their hidden
//                        activations with the forward's f16x2 arithmetic (mlp_f2_heads.h; one scale per tile instead of one per
//                        sample) -- twice per layer, because an MFMA's output orientation is a matter of which operand is which:
//                            D = W (A) x H_l (B)    ->  lane = sample,  registers = features : the next layer's B operand (as in the forward)
//                            D = H_l (A) x W (B)    ->  lane = feature, registers = samples  : H_{l+1} exactly as the weight gradient's
//                                                       MFMA wants it (reduction index = sample), same registers as the first form
//                        so the sample <-> feature transposition that the workspace rows performed through HBM costs six more fp16
//                        MFMAs per k block and nothing else.  The G rows arrive from the workspace in that same layout (lane =
//                        feature row, 16 B pieces).  Products G x H as exact three-way bf16 splits (b3_device.h: no scales, fp32
//                        accumulate in registers for the whole launch, one full-line atomic flush); the <= 4-row output layers on the VALU.
//
// Both kernels are wave-private: LDS holds the two heads' weights only.
#include "mlp_stage.h"
#include "mlp_f2_heads.h"
#include "b3_device.h"
#include <algorithm>

// phase boundaries of wgrad_rc_kernel's tile loop.  Scheduling barriers there cost 15 spilled registers and left 200-instruction VALU blocks
// between bare MFMA runs; without them hipcc interleaves the conversions with the MFMAs (a lone wave overlaps the two pipes only inside its
// own instruction stream) and allocates 250 + 240 registers without a spill.  -DTN_RC_SB='__builtin_amdgcn_sched_barrier(0);' restores them.
#ifndef TN_RC_SB
#define TN_RC_SB
#endif
#ifdef TN_PHASE_TIMERS            // dev build: where a tile's cycles go (scripts/phase_time_rc.py)
__device__ unsigned long long tn_phase_cycles_rc[16];
#define TN_PTR_BEGIN unsigned long long ptr_ = __builtin_amdgcn_s_memtime();
#define TN_PTR(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (tn::lane_id() == 0) atomicAdd(&tn_phase_cycles_rc[k], n_ - ptr_); ptr_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int tn_debug_phase_cycles_rc(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(tn_phase_cycles_rc), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(tn_phase_cycles_rc), z, sizeof(z)); }
    return 0;
}
#else
#define TN_PTR_BEGIN
#define TN_PTR(k)
#endif
namespace {

using tn::f32x16;
using tn::f32x4;
using namespace tn::mlp;
namespace b3 = tn::b3;

constexpr int H = 64, NH = 4;

// maximum over the wave, in every lane, without the LDS crossbar (six dependent ds_bpermute round trips, five times per tile on a wave
// that runs alone on its SIMD): DPP within a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)));
}
__device__ __forceinline__ float wave_max(float m) {
    m = dpp_max<0xB1>(m);            // quad_perm [1, 0, 3, 2]
    m = dpp_max<0x4E>(m);            // quad_perm [2, 3, 0, 1]
    m = dpp_max<0x141>(m);           // row_half_mirror
    m = dpp_max<0x128>(m);           // row_ror:8
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}

// 16 values of row `i` of a [32 features][32 samples] row block in D-register order: register r <-> sample (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ void load_block_f(const float *__restrict__ rows, int i, int h, float (&v)[16]) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(rows + i * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = p[2 * q];
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
}

__device__ __forceinline__ void split_block(const float (&v)[16], b3::Op (&op)[2]) {
    const float a[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    const float b[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    op[0] = b3::split8(a);
    op[1] = b3::split8(b);
}

// one k block of a layer in one or both orientations (see the header): accS[ob] = W x (lane = sample), accF[ob] = x^T W^T (lane = feature).
// The weight operands are loaded by load_w() one k block AHEAD of their MFMAs (a wave runs alone on its SIMD: nobody else covers
// the LDS round trip).
struct WOp { u32x4h a0h, a0l, a1h, a1l; };
__device__ __forceinline__ WOp load_w(const _Float16 *__restrict__ Wh, int plane, int stride, int j, int h, int b) {
    const _Float16 *p0 = Wh + j * stride + 16 * b + 8 * h, *p1 = p0 + 32 * stride;
    WOp w;
    w.a0h = *reinterpret_cast<const u32x4h *>(p0); w.a0l = *reinterpret_cast<const u32x4h *>(p0 + plane);
    w.a1h = *reinterpret_cast<const u32x4h *>(p1); w.a1l = *reinterpret_cast<const u32x4h *>(p1 + plane);
    return w;
}
// `first`: the k block that opens an accumulation -- its MFMAs take the inline constant 0 as C operand instead of a zeroed register tile
// (16 v_accvgpr_write per tile and chain: 300 instructions per 32 samples)
template <bool S>
__device__ __forceinline__ void mma_sf(const WOp &w, const u32x4h &bh, const u32x4h &bl, f32x16 (&accS)[2], f32x16 (&accF)[2], bool first = false) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (S) {
        accS[0] = mfma_f16(w.a0l, bh, first ? z : accS[0]);
        accS[1] = mfma_f16(w.a1l, bh, first ? z : accS[1]);
    }
    accF[0] = mfma_f16(bh, w.a0l, first ? z : accF[0]);
    accF[1] = mfma_f16(bh, w.a1l, first ? z : accF[1]);
    if constexpr (S) {
        accS[0] = mfma_f16(w.a0h, bl, accS[0]);
        accS[1] = mfma_f16(w.a1h, bl, accS[1]);
    }
    accF[0] = mfma_f16(bl, w.a0h, accF[0]);
    accF[1] = mfma_f16(bl, w.a1h, accF[1]);
    if constexpr (S) {
        accS[0] = mfma_f16(w.a0h, bh, accS[0]);
        accS[1] = mfma_f16(w.a1h, bh, accS[1]);
    }
    accF[0] = mfma_f16(bh, w.a0h, accF[0]);
    accF[1] = mfma_f16(bh, w.a1h, accF[1]);
}

__device__ __forceinline__ void zero2(f32x16 (&a)[2]) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ob][r] = 0.0f;
}

struct RcArgs {
    MlpArgs a, b;                 // colour head (5 layers) and sigma head (2 layers) in their f16x2 LDS form (plan_f2)
    const float *x, *aux;         // feature rows [n, 96]; per-ray table of TN_ENC_AUX_CAT
    const float *ws_a, *ws_b;     // the heads' workspaces behind the chain: G rows and g_pre rows
    float *gW[5], *gB[5];         // colour head: layers 1 .. 4 are this kernel's
    float *gWs, *gBs;             // sigma head: layer 1
    int64_t n;
};

// compile-time shape (the reference's decoders, src/run.py:133-139: colour head 3 outputs on [PE_8(d) (48), d (3), x (96)], sigma head 1):
// OA / OB = output rows of the two heads, NGA / NBA = groups of 8 / k blocks of 16 table columns
// (The body is a function of __restrict__ pointers: no store of the kernel aliases a load.)  Tried for the G rows and measured, not kept:
// LDS-direct requests (global_load_lds_dwordx4 into a wave-private 8 KB buffer: no staging registers).  Through the builtin hipcc's waitcnt
// pass serialises them completely -- vmcnt(0) in front of every request and of every ds_read behind one: 0.64 ms against 0.54 --, through
// inline assembly the requests overlap, but the 16 address registers per stage brought 19 spills back, and a scratch reload drains the
// queue like any other vmcnt(0).  The register form below (two sets in turn, requested a stage + a layer ahead) has none.
// Also measured, not kept: the weight gradients dealt out by layer to FOUR launches with at most four accumulator tiles per wave (256
// registers, two waves per SIMD, no spills; each role re-evaluates the part of the chain it needs: 564 MFMAs per tile instead of 420):
// 0.21 + 0.25 + 0.26 + 0.29 = 1.00 ms against this kernel's 0.54 -- what a role saves in matrix work it pays several times over in the
// per-tile work every role repeats (feature rows, table rows, their conversion, the first layer).
template <int WPB, bool AUX, int OA, int OB, int NGA, int NBA>
__device__ __forceinline__ void wgrad_rc_body(const RcArgs &p, const float *__restrict__ ldsa, const float *__restrict__ ldsb, float *__restrict__ /*unused*/,
                                              const float *__restrict__ px, const float *__restrict__ paux, const int *__restrict__ paidx,
                                              const float *__restrict__ pws_a, const float *__restrict__ pws_b)
{
    const int lane = tn::lane_id(), j_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = p.n, n_tiles = (n + 31) >> 5;
    const int Rta = stash_rows(H, NH, 0), Rtb = stash_rows(H, 1, 0);
    constexpr int nga = NGA, nba = NBA;
    constexpr bool has_aux = AUX;

    f32x16 acc[3][2][2];          // dW_{l+1}: [layer][row block tn][column block tk]
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int tn_ = 0; tn_ < 2; ++tn_)
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[l][tn_][tk][r] = 0.0f;
    float db[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float dwo[OA][2], dws[OB][2], dbo[OA], dbs[OB];
#pragma unroll
    for (int o = 0; o < OA; ++o) { dwo[o][0] = dwo[o][1] = 0.f; dbo[o] = 0.f; }
#pragma unroll
    for (int o = 0; o < OB; ++o) { dws[o][0] = dws[o][1] = 0.f; dbs[o] = 0.f; }

    // The tile's inputs -- feature rows and per-ray table rows, lane = sample as in the forward -- are requested one tile ahead
    // (a wave runs alone on its SIMD: nobody else hides an HBM round trip), behind the first layer, where their registers are free.
    f32x4 fr[12], av[8];
    auto fetch_x = [&](int64_t t) {
        int64_t r = t * 32 + j_;
        r = r < n ? r : n - 1;
        const float *xr = px + r * 96 + 4 * h_;
#pragma unroll
        for (int g = 0; g < 12; ++g) fr[g] = *reinterpret_cast<const f32x4 *>(xr + 8 * g);
    };
    // table row of a sample: its index (an unconditional load: a load under `aux_index ? ... : ...` is a blocking load) is
    // requested at the top of the tile BEFORE the one that gathers through it
    int aidx = 0;
    auto fetch_aidx = [&](int64_t t) {
        int64_t r = t * 32 + j_;
        r = r < n ? r : n - 1;
        if constexpr (has_aux) aidx = paidx[r];
    };
    auto fetch_aux = [&]() {
        if constexpr (has_aux) {
            const float *arow = paux + (int64_t)aidx * p.a.aux_stride + 4 * h_;
#pragma unroll
            for (int g = 0; g < 8; ++g) av[g] = *reinterpret_cast<const f32x4 *>(arow + 8 * (g < nga ? g : nga - 1));
        }
    };
    {
        const int64_t t0 = (int64_t)blockIdx.x * WPB + wave;
        fetch_aidx(t0 < n_tiles ? t0 : n_tiles - 1);
        fetch_aux();
        fetch_x(t0 < n_tiles ? t0 : n_tiles - 1);
    }
    // G rows of the weight-gradient stages: two register sets in turn, each requested a whole stage + layer ahead of its use
    float gqa[2][16], gqb[2][16];
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        int j = j_, h = h_;
        asm volatile("" : "+v"(j), "+v"(h));
        const float *wsa = pws_a + tile * (int64_t)(Rta * 32);
        const float *wsb = pws_b + tile * (int64_t)(Rtb * 32);
        const _Float16 *W0 = reinterpret_cast<const _Float16 *>(ldsa + p.a.w_off[0]);
        const int plane0 = p.a.f2_plane[0], st0 = p.a.stride[0];
        const int64_t tile_next = tile + (int64_t)gridDim.x * WPB < n_tiles ? tile + (int64_t)gridDim.x * WPB : n_tiles - 1;
        fetch_aidx(tile_next);
        TN_PTR_BEGIN

        f32x16 accS[2], accF[2], actS[2], actF[2];
        // ---- colour head, layer 0, table columns first (their registers are the first to go): both orientations ----
        if constexpr (has_aux) {
            WOp w = load_w(W0, plane0, st0, j, h, 6);
            float m = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) m = fmaxf(fmaxf(m, fmaxf(fabsf(av[g][0]), fabsf(av[g][1]))), fmaxf(fabsf(av[g][2]), fabsf(av[g][3])));
            float s_a, inv_a;
            f2_scales(wave_max(m), s_a, inv_a);
#pragma unroll
            for (int ba = 0; ba < nba; ++ba) {
                const WOp wn = load_w(W0, plane0, st0, j, h, 6 + (ba + 1 < nba ? ba + 1 : ba));
                const float v[8] = {av[2 * ba][0], av[2 * ba][1], av[2 * ba][2], av[2 * ba][3], av[2 * ba + 1][0], av[2 * ba + 1][1], av[2 * ba + 1][2], av[2 * ba + 1][3]};
                u32x4h bh, bl;
                f2_split8(v, s_a, bh, bl);
                mma_sf<true>(w, bh, bl, accS, accF, ba == 0);
                TN_RC_SB
                w = wn;
            }
            const float ca = inv_a * ldsa[p.a.f2_scale + 1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { actS[ob][r] = accS[ob][r] * ca; actF[ob][r] = accF[ob][r] * ca; }
            }
        } else {
            zero2(actS); zero2(actF);
        }
        TN_RC_SB
        TN_PTR(0)
        // ---- the feature rows as fp16 operands under one scale per tile ----
        u32x4h xbh[6], xbl[6];
        float inv_x;
        {
            float m = 0.0f;
#pragma unroll
            for (int g = 0; g < 12; ++g) m = fmaxf(fmaxf(m, fmaxf(fabsf(fr[g][0]), fabsf(fr[g][1]))), fmaxf(fabsf(fr[g][2]), fabsf(fr[g][3])));
            float s_x;
            f2_scales(wave_max(m), s_x, inv_x);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const float v[8] = {fr[2 * b][0], fr[2 * b][1], fr[2 * b][2], fr[2 * b][3], fr[2 * b + 1][0], fr[2 * b + 1][1], fr[2 * b + 1][2], fr[2 * b + 1][3]};
                f2_split8(v, s_x, xbh[b], xbl[b]);
            }
        }
        TN_RC_SB
        TN_PTR(1)
        // G_1 rows: needed behind the first layer, requested in front of it
        load_block_f(wsa + (int64_t)((NH + 1) * H) * 32, j, h, gqa[0]);
        load_block_f(wsa + (int64_t)((NH + 1) * H + 32) * 32, j, h, gqa[1]);
        // ---- colour head, layer 0, x columns ----
        {
            WOp w = load_w(W0, plane0, st0, j, h, 0);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const WOp wn = load_w(W0, plane0, st0, j, h, b + 1 < 6 ? b + 1 : b);
                mma_sf<true>(w, xbh[b], xbl[b], accS, accF, b == 0);
                TN_RC_SB
                w = wn;
            }
            const float cx = inv_x * ldsa[p.a.f2_scale + 1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f32x16 bias = tn::bias_tile(ldsa + p.a.b_off[0], ob, h);
                const float bl_ = ldsa[p.a.b_off[0] + 32 * ob + j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    actS[ob][r] = fmaf(accS[ob][r], cx, actS[ob][r]) + bias[r];
                    actF[ob][r] = fmaf(accF[ob][r], cx, actF[ob][r]) + bl_;
                }
                actS[ob] = tn::relu16(actS[ob]);
                actF[ob] = tn::relu16(actF[ob]);
            }
        }
        TN_RC_SB
        TN_PTR(2)
        // ---- sigma head: H_1s (lane = feature) -> dW_1s, db_1s on the VALU (<= 4 output rows) ----
        {
            const _Float16 *W0s = reinterpret_cast<const _Float16 *>(ldsb + p.b.w_off[0]);
            const int pls = p.b.f2_plane[0], sts = p.b.stride[0];
            float gs[OB][16];                                     // g_pre rows of the sigma head: requested in front of its MFMAs
#pragma unroll
            for (int o = 0; o < OB; ++o) load_block_f(wsb + 2 * H * 32, o, h, gs[o]);
            WOp w = load_w(W0s, pls, sts, j, h, 0);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const WOp wn = load_w(W0s, pls, sts, j, h, b + 1 < 6 ? b + 1 : b);
                mma_sf<false>(w, xbh[b], xbl[b], accS, accF, b == 0);
                TN_RC_SB
                w = wn;
            }
            const float c = inv_x * ldsb[p.b.f2_scale + 1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const float bias = ldsb[p.b.b_off[0] + 32 * ob + j];
#pragma unroll
                for (int r = 0; r < 16; ++r) accF[ob][r] = fmaf(accF[ob][r], c, bias);
                accF[ob] = tn::relu16(accF[ob]);
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) {
                float s0 = 0.f, s1 = 0.f, sg = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0 = fmaf(gs[o][r], accF[0][r], s0); s1 = fmaf(gs[o][r], accF[1][r], s1); sg += gs[o][r]; }
                dws[o][0] += s0; dws[o][1] += s1; dbs[o] += sg;
            }
        }
        TN_RC_SB
        TN_PTR(3)

        // ---- hidden layers: dW_l += G_l H_l^T with H_l = actF (lane = feature), then H_{l+1} in the orientation(s) still needed ----
        float go[OA][16];               // g_pre rows of the colour head (requested in front of the last layer)
#pragma unroll
        for (int l = 1; l <= 3; ++l) {
            const _Float16 *Wl = reinterpret_cast<const _Float16 *>(ldsa + p.a.w_off[l]);
            const int plane = p.a.f2_plane[l], stride = p.a.stride[l];
            // the stage's G rows: out of the wave's LDS buffer (l = 1, 3) or the registers requested a layer and a half ago (l = 2); the
            // requests that follow are in flight for 12 k cycles or more
            // this stage's rows are in one register set; the next stage's are requested into the other one right here
            float (&gq)[2][16] = (l == 2) ? gqb : gqa;
            if (l < 3) {
                float (&gn)[2][16] = (l == 1) ? gqb : gqa;
                load_block_f(wsa + (int64_t)((NH + l + 1) * H) * 32, j, h, gn[0]);
                load_block_f(wsa + (int64_t)((NH + l + 1) * H + 32) * 32, j, h, gn[1]);
            }
            // both operands as bf16 triplets: H_l from registers, G_l (lane = feature row, register = sample)
            {
                b3::Op HB[2][2];
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = actF[tk][r];
                    split_block(v, HB[tk]);
                }
#pragma unroll
                for (int tn_ = 0; tn_ < 2; ++tn_) {
                    float sg = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sg += gq[tn_][r];
                    db[l - 1][tn_] += sg;
                    b3::Op GA[2];
                    split_block(gq[tn_], GA);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int tk = 0; tk < 2; ++tk) acc[l - 1][tn_][tk] = b3::mfma6(GA[c], HB[tk][c], acc[l - 1][tn_][tk]);
                    TN_RC_SB
                }
            }
            TN_PTR(3 + l)
            // register prefetches behind the stage's MFMAs, where the operand triplets have gone: the table rows of the next tile (32
            // registers across the hidden layers), and in the last layer its feature rows (48) and this tile's g_pre rows
            if (l == 1) fetch_aux();
            if (l == 3) {
#pragma unroll
                for (int o = 0; o < OA; ++o) load_block_f(wsa + 2 * NH * H * 32, o, h, go[o]);
                fetch_x(tile_next);
            }
            // layer l: H_l (actS, lane = sample, all >= 0) -> H_{l+1}
            WOp w = load_w(Wl, plane, stride, j, h, 0);
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, actS[kb][r]);
            float s, inv;
            f2_scales(wave_max(m), s, inv);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const WOp wn = load_w(Wl, plane, stride, j, h, b + 1 < 4 ? b + 1 : b);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = actS[b >> 1][8 * (b & 1) + e];
                u32x4h bh, bl;
                f2_split8(v, s, bh, bl);
                if (l < 3) mma_sf<true>(w, bh, bl, accS, accF, b == 0);
                else mma_sf<false>(w, bh, bl, accS, accF, b == 0);       // (H_4 only feeds the output layer's gradient)
                TN_RC_SB
                w = wn;
            }
            const float c = inv * ldsa[p.a.f2_scale + 2 * l + 1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const float bl_ = ldsa[p.a.b_off[l] + 32 * ob + j];
#pragma unroll
                for (int r = 0; r < 16; ++r) actF[ob][r] = fmaf(accF[ob][r], c, bl_);
                actF[ob] = tn::relu16(actF[ob]);
                if (l < 3) {
                    const f32x16 bias = tn::bias_tile(ldsa + p.a.b_off[l], ob, h);
#pragma unroll
                    for (int r = 0; r < 16; ++r) actS[ob][r] = fmaf(accS[ob][r], c, bias[r]);
                    actS[ob] = tn::relu16(actS[ob]);
                }
            }
            TN_RC_SB
            TN_PTR(6 + l)
        }
        // ---- output layer of the colour head: dW_4 = g_pre H_4^T, db_4 on the VALU ----
#pragma unroll
        for (int o = 0; o < OA; ++o) {
            float s0 = 0.f, s1 = 0.f, sg = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0 = fmaf(go[o][r], actF[0][r], s0); s1 = fmaf(go[o][r], actF[1][r], s1); sg += go[o][r]; }
            dwo[o][0] += s0; dwo[o][1] += s1; dbo[o] += sg;
        }
        TN_PTR(10)
    }

    // ---- flush: D[row = G feature][column = H feature], lanes = consecutive columns of one weight row ----
    const int i = j_, h = h_;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int tn_ = 0; tn_ < 2; ++tn_) {
#pragma unroll
            for (int tk = 0; tk < 2; ++tk) {
                tn::pin16(acc[l][tn_][tk]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = 32 * tn_ + (r & 3) + 8 * (r >> 2) + 4 * h;
                    atomicAdd(&p.gW[l + 1][nn * H + 32 * tk + i], acc[l][tn_][tk][r]);
                }
            }
            float s = db[l][tn_];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&p.gB[l + 1][32 * tn_ + i], s);
        }
    }
#pragma unroll
    for (int o = 0; o < OA; ++o) {
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
            float s = dwo[o][tk];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&p.gW[4][o * H + 32 * tk + i], s);
        }
        float s = dbo[o];
        s += __shfl_xor(s, 32, 64);
        if (lane == 0) atomicAdd(&p.gB[4][o], s);
    }
#pragma unroll
    for (int o = 0; o < OB; ++o) {
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
            float s = dws[o][tk];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&p.gWs[o * H + 32 * tk + i], s);
        }
        float s = dbs[o];
        s += __shfl_xor(s, 32, 64);
        if (lane == 0) atomicAdd(&p.gBs[o], s);
    }
}

template <int WPB, bool AUX, int OA, int OB, int NGA, int NBA>
__global__ __launch_bounds__(WPB * 64) void wgrad_rc_kernel(RcArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights_f2(p.a, lds);
    stage_weights_f2(p.b, lds + p.a.lds_floats);
    wgrad_rc_body<WPB, AUX, OA, OB, NGA, NBA>(p, lds, lds + p.a.lds_floats, nullptr, p.x, p.aux,
                                              p.a.aux_index, p.ws_a, p.ws_b);
}

// ------------------------------------------------------------------------------------------------
// first layers of both heads
// ------------------------------------------------------------------------------------------------
// dW_0 = G_0 [x | table]^T of the colour head (64 x 147) and dW_0 = G_0 x^T of the sigma head (64 x 96), one launch.  The first draft read
// both MFMA operands straight from global memory in operand layout (lane = row, 16 B pieces 128 B apart: 64 requests per load
// instruction, and G_0 once per column block): 0.56 ms, bound by the texture-address rate.  This form is the paired tiling of
// mlp_bwd2.hip's mlp_wgrad4_kernel reduced to its first-layer waves: the tile's G_0 rows of both heads, its x rows and its table rows
// are staged ONCE per workgroup in LDS (coalesced 16-byte chunks, two buffers, one barrier per tile, every load a full iteration ahead),
// wave w < 5 owns column block w of the colour head (tiles tn = 0, 1), waves 5 .. 7 column blocks 0 .. 2 of the sigma head.
// Measured and not kept: the same tiling with every element split into its bf16 triplet on the way from the staging registers to LDS
// (x / table rows stored transposed, 24 bf16 MFMAs per wave and tile instead of 32 fp32 ones): 0.43 ms against this form's 0.36 -- twelve
// 2-byte LDS writes per x chunk and one workgroup per CU (139 KB) cost more than the matrix pipe gains.
struct FlArgs {
    const float *x, *aux;         // [n, 96]; per-ray table [*, aux_stride]
    const int *aux_index;
    int aux_stride, pe, K0_a, K0_b;              // pe = table columns in use = torch columns 0 .. pe-1 of the colour head's first layer
    const float *ws_a, *ws_b;
    int rt_a, rt_b, g0_a, g0_b;   // rows per tile; first G_0 row
    float *gW0, *gB0, *gW0s, *gB0s;
    int64_t n;
};

constexpr int FL_RS = 36;                        // LDS row stride (floats): conflict-free ds_read_b128 across 16 lanes
constexpr int FL_IN = 96, FL_AW = 56;            // x columns, staged table columns
constexpr int FL_BUF = 128 * FL_RS + 32 * FL_IN + 32 * FL_AW;        // floats per buffer

__global__ __launch_bounds__(512) void wgrad_first_kernel(FlArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 8, NCH = 4;
    constexpr int g_chunks = 128 * 8, x_chunks = 32 * FL_IN / 4;
    const int lane = tn::lane_id(), i_ = lane & 31, h_ = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = p.n, n_tiles = (n + 31) >> 5;
    const bool colour = wave < 5;                 // (wave-uniform)
    const int tk = colour ? wave : wave - 5;
    f32x16 acc[2];
    float dbacc[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    f32x4 pre[NCH];
    f32x4 preA = {0.f, 0.f, 0.f, 0.f};
    int aidx = 0;
    // every load unconditional from clamped addresses (see mlp_bwd2.hip): chunk c < 512: G_0 rows of the colour head, < 1024: of the sigma
    // head, then the tile's x rows (contiguous)
    auto prefetch = [&](int64_t tile, int64_t tile_after) {
        const f32x4 *sa = reinterpret_cast<const f32x4 *>(p.ws_a + tile * (int64_t)p.rt_a * 32 + (int64_t)p.g0_a * 32);
        const f32x4 *sb = reinterpret_cast<const f32x4 *>(p.ws_b + tile * (int64_t)p.rt_b * 32 + (int64_t)p.g0_b * 32);
        const int64_t x0 = tile * 32 * (int64_t)FL_IN;
        const int64_t xlast = n * (int64_t)FL_IN - 4;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            int64_t e = x0 + 4 * (int64_t)(c - g_chunks);
            e = e < 0 ? 0 : (e > xlast ? xlast : e);
            const f32x4 *px = reinterpret_cast<const f32x4 *>(p.x + e);
            const f32x4 *pg = c < 512 ? sa + c : sb + ((c - 512) & 511);
            pre[k] = *(c < g_chunks ? pg : px);
        }
        {
            const int s_ = (threadIdx.x >> 4) & 31, part = threadIdx.x & 15;
            preA = *reinterpret_cast<const f32x4 *>(p.aux + (int64_t)aidx * p.aux_stride + (4 * part < FL_AW ? 4 * part : 0));
            int64_t r_ = tile_after * 32 + s_;
            r_ = r_ < n ? r_ : n - 1;
            aidx = p.aux_index[r_];
        }
    };
    float *dummy = lds + 2 * FL_BUF;
    auto commit = [&](float *buf) {
        float *ldsX = buf + 128 * FL_RS, *ldsA = ldsX + 32 * FL_IN;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = threadIdx.x + k * NW * 64;
            float *dst = c < g_chunks ? buf + (c >> 3) * FL_RS + (c & 7) * 4 : (c < g_chunks + x_chunks ? ldsX + 4 * (c - g_chunks) : dummy);
            *reinterpret_cast<f32x4 *>(dst) = pre[k];
        }
        const int s_ = threadIdx.x >> 4, part = threadIdx.x & 15;
        float *dst = 4 * part < FL_AW ? ldsA + s_ * FL_AW + 4 * part : dummy;
        *reinterpret_cast<f32x4 *>(dst) = preA;
    };
    const int64_t last = n_tiles - 1;
    int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    {
        int64_t r_ = tile * 32 + ((threadIdx.x >> 4) & 31);
        r_ = r_ < n ? r_ : n - 1;
        aidx = p.aux_index[r_];
    }
    {
        const int64_t t1 = tile + gridDim.x, t2 = t1 + gridDim.x;
        prefetch(tile, t1 < last ? t1 : last);
        commit(lds);
        __syncthreads();
        prefetch(t1 < last ? t1 : last, t2 < last ? t2 : last);
    }
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        int i = i_, h = h_;
        asm volatile("" : "+v"(i), "+v"(h));
        float *buf = lds + cur * FL_BUF;
        float *nbuf = lds + (cur ^ 1) * FL_BUF;
        const int64_t t2 = tile + 2 * (int64_t)gridDim.x, t3 = t2 + gridDim.x;
        commit(nbuf);
        prefetch(t2 < last ? t2 : last, t3 < last ? t3 : last);
        const float *ldsX = buf + 128 * FL_RS, *ldsA = ldsX + 32 * FL_IN;
        {
            const int grow = colour ? 0 : 64;
            const f32x4 *g0 = reinterpret_cast<const f32x4 *>(buf + (grow + i) * FL_RS + 16 * h);
            const f32x4 *g1 = reinterpret_cast<const f32x4 *>(buf + (grow + 32 + i) * FL_RS + 16 * h);
            // B operand: column 32 tk + i of the sample-major x rows, or (colour head, tk >= 3) of the table rows
            const int q = 32 * tk + i;
            const bool xcol = q < FL_IN;
            const int ac = q - FL_IN < FL_AW ? q - FL_IN : FL_AW - 1;            // (columns >= pe are never flushed)
            const float *cp = xcol ? ldsX + 16 * h * FL_IN + q : ldsA + 16 * h * FL_AW + (ac < 0 ? 0 : ac);
            const int cstride = xcol ? FL_IN : FL_AW;
            float gs0 = 0.f, gs1 = 0.f;
            // products as exact bf16 triplets, split by the wave that multiplies them (k block c = the lane's samples 8 c .. 8 c + 7 of its
            // half): 24 bf16 MFMAs + ~270 VALU instructions per wave and tile instead of 32 fp32 MFMAs (2 048 matrix-pipe cycles)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 ga = g0[2 * c], gb = g0[2 * c + 1], gc = g1[2 * c], gd = g1[2 * c + 1];
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = cp[(8 * c + u) * cstride];
                const float v0[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
                const float v1[8] = {gc[0], gc[1], gc[2], gc[3], gd[0], gd[1], gd[2], gd[3]};
                const b3::Op XB = b3::split8(xv);
                acc[0] = b3::mfma6(b3::split8(v0), XB, acc[0]);
                acc[1] = b3::mfma6(b3::split8(v1), XB, acc[1]);
                gs0 += ((ga[0] + ga[1]) + (ga[2] + ga[3])) + ((gb[0] + gb[1]) + (gb[2] + gb[3]));
                gs1 += ((gc[0] + gc[1]) + (gc[2] + gc[3])) + ((gd[0] + gd[1]) + (gd[2] + gd[3]));
            }
            if (tk == 0) { dbacc[0] += gs0; dbacc[1] += gs1; }
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- flush: lanes = consecutive columns of one weight row ----
    const int i = i_, h = h_;
    const int q = 32 * tk + i;
    int col; bool ok;
    float *gW; int ldw;
    if (colour) { gW = p.gW0; ldw = p.K0_a; ok = q < FL_IN || q - FL_IN < p.pe; col = q < FL_IN ? p.pe + q : q - FL_IN; }     // torch order: [table, x]
    else { gW = p.gW0s; ldw = p.K0_b; ok = true; col = q; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        tn::pin16(acc[m]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ok) atomicAdd(&gW[(int64_t)nn * ldw + col], acc[m][r]);
        }
        if (tk == 0) {
            float s = dbacc[m];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&(colour ? p.gB0 : p.gB0s)[32 * m + i], s);
        }
    }
}

}  // namespace

// 1 when tn_mlp_bwd_pair / tn_kplanes_mlp_bwd_pair can run `desc` / `partner` under TN_MLP_LEAN (callers decide the forward's form with it)
extern "C" int tn_mlp_lean_supported(const tn_mlp_desc *desc, const tn_mlp_desc *partner)
{
    if (!desc || !partner) return 0;
    const bool f2 = (desc->flags & TN_MLP_F16X2) && (partner->flags & TN_MLP_F16X2);
    return f2 && desc->n_layers == 5 && partner->n_layers == 2 && desc->in_dim == 96 && partner->in_dim == 96 && desc->encoding == TN_ENC_AUX_CAT &&
           partner->encoding == TN_ENC_NONE && desc->dims[0] == 147 && desc->dims[1] == 64 && desc->dims[2] == 64 && desc->dims[3] == 64 &&
           desc->dims[4] == 64 && desc->dims[5] == 3 && partner->dims[0] == 96 && partner->dims[1] == 64 && partner->dims[2] == 1 &&
           desc->aux_stride >= 56 && (desc->aux_stride & 3) == 0;
}

// Weight / bias gradients of the paired heads under TN_MLP_LEAN: `desc` = the 5-layer head (TN_ENC_AUX_CAT or TN_ENC_NONE on 96 inputs),
// `partner` = the 2-layer head on the same x; workspaces as the chain half of tn_mlp_bwd_pair left them.  Gradients accumulate (+=).
extern "C" int tn_mlp_wgrad_lean_pair(const tn_mlp_desc *desc, const tn_mlp_desc *partner, const float *x, const float *aux, int64_t n,
                                      float *const *gw, float *const *gb, float *const *gws, float *const *gbs, const float *ws_a,
                                      const float *ws_b, void *stream)
{
    TN_REQUIRE(desc && partner && x && gw && gb && gws && gbs && ws_a && ws_b, TN_E_NULL, "tn_mlp_wgrad_lean_pair: null pointer");
    RcArgs p;
    int Ha = 0, Hb = 0;
    if (int rc = plan(desc, p.a, Ha)) return rc;
    if (int rc = plan(partner, p.b, Hb)) return rc;
    TN_REQUIRE(Ha == 64 && Hb == 64 && p.a.n_layers == 5 && p.b.n_layers == 2 && p.a.in_dim == 96 && p.b.in_dim == 96 && p.a.out_dim <= 4 &&
                   p.b.out_dim <= 4 && p.a.enc == TN_ENC_AUX_CAT && p.b.enc == TN_ENC_NONE && p.a.f2 && p.b.f2 &&
                   f2_head_ok(p.a, 64) && f2_head_ok(p.b, 64) && p.a.out_dim == 3 && p.b.out_dim == 1 && p.a.K0_pad - p.a.in_dim == 56,
               TN_E_CONFIG, "TN_MLP_LEAN: the reference's decoders (run.py:133-139) as f16x2 heads (TN_MLP_F16X2): colour 147 -> 64 x 4 -> 3 on "
                            "[per-ray table (51), x (96)], sigma 96 -> 64 -> 1");
    TN_REQUIRE(aux && ((uintptr_t)aux & 15) == 0 && p.a.aux_index, TN_E_NULL, "TN_MLP_LEAN: aux table (16-byte aligned) and aux_index are required");
    TN_REQUIRE((((uintptr_t)x | (uintptr_t)ws_a | (uintptr_t)ws_b) & 15) == 0, TN_E_ALIGN, "TN_MLP_LEAN: x / workspaces must be 16-byte aligned");
    if (n <= 0) return TN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_tiles = (n + 31) / 32;
    // ---- first layers ----
    {
        FlArgs f;
        f.x = x; f.aux = aux; f.aux_index = p.a.aux_index; f.aux_stride = p.a.aux_stride;
        f.pe = p.a.K0 - p.a.in_dim;
        f.ws_a = ws_a; f.ws_b = ws_b;
        f.rt_a = stash_rows(H, NH, 0); f.rt_b = stash_rows(H, 1, 0);
        f.g0_a = NH * H; f.g0_b = H;
        f.K0_a = p.a.K0; f.K0_b = p.b.K0;
        f.gW0 = gw[0]; f.gB0 = gb[0]; f.gW0s = gws[0]; f.gB0s = gbs[0];
        f.n = n;
        TN_REQUIRE(f.gW0 && f.gB0 && f.gW0s && f.gB0s, TN_E_NULL, "TN_MLP_LEAN: null gradient pointer");
        TN_REQUIRE(f.aux_stride >= FL_AW && f.pe <= FL_AW, TN_E_CONFIG, "TN_MLP_LEAN: table rows of 56 .. floats, at most 56 columns in use");
        const size_t lds1 = (2 * (size_t)FL_BUF + 16) * 4;
        hipError_t e1 = hipFuncSetAttribute((const void *)wgrad_first_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e1 != hipSuccess) { tn::set_error("TN_MLP_LEAN: cannot reserve %zu B of LDS: %s", lds1, hipGetErrorString(e1)); return (int)e1; }
        const int64_t blocks = std::min<int64_t>(n_tiles, 256 * 2);
        wgrad_first_kernel<<<dim3((unsigned)blocks), dim3(512), lds1, s>>>(f);
        if (int rc = tn::check_launch("wgrad_first_kernel")) return rc;
    }
    // ---- hidden and output layers, activations rebuilt ----
    plan_f2(p.a, 64);
    plan_f2(p.b, 64);
    p.x = x; p.aux = aux; p.ws_a = ws_a; p.ws_b = ws_b; p.n = n;
    for (int l = 0; l < 5; ++l) { p.gW[l] = gw[l]; p.gB[l] = gb[l]; TN_REQUIRE(gw[l] && gb[l], TN_E_NULL, "TN_MLP_LEAN: null gradient pointer"); }
    p.gWs = gws[1]; p.gBs = gbs[1];
    TN_REQUIRE(p.gWs && p.gBs, TN_E_NULL, "TN_MLP_LEAN: null gradient pointer");
    constexpr int WPB = 4;                    // one wave per SIMD: 192 accumulator registers per wave
    const size_t lds_bytes = ((size_t)p.a.lds_floats + (size_t)p.b.lds_floats) * 4;
    TN_REQUIRE(lds_bytes <= (size_t)LDS_LIMIT_BYTES, TN_E_CONFIG, "TN_MLP_LEAN: weights do not fit LDS");
    auto kern = wgrad_rc_kernel<WPB, true, 3, 1, 7, 4>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("TN_MLP_LEAN: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t blocks = std::min<int64_t>((n_tiles + WPB - 1) / WPB, 256);
    kern<<<dim3((unsigned)blocks), dim3(WPB * 64), lds_bytes, s>>>(p);
    return tn::check_launch("wgrad_rc_kernel");
}
