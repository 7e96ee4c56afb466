// K-Planes backward for ONE scale and one wave's 32 samples: transposed, run-merged scatter (see kplanes.hip for the
// measurements behind it).  Shared by the stand-alone kernel (kplanes.hip) and by the MLP data-gradient chain, which
// hands its d(loss)/d(features) registers straight to it (mlp_bwd2.hip, tn_kplanes_mlp_bwd_pair).
//
// Measured on MI355X (scripts/microbench/atomic_patterns.hip): a wave64 global_atomic_add_f32 whose lanes hit 64 different
// cache lines retires ~20 G lane-atomics/s, one whose half-waves each cover the 32 consecutive dwords of ONE line ~270 G/s.
// The forward mapping (lane = sample) is the slow pattern, so the scatter is transposed through a 5.5 KiB per-wave LDS
// tile: phase A (lane = sample, channel group) writes the 32x32 tile of d(feat)/d(plane value) plus the 4 tap offsets /
// weights of every sample; phase B (lane = channel) walks the samples in order -- consecutive samples of a ray fall into
// the same cell for several steps (a straight line visits the cells of a plane monotonically), so the contributions of a
// RUN of samples are accumulated in registers and ONE full-line atomic per tap is issued at the end of the run.  Run
// boundaries depend on the cell only, so they are the same for all four taps: phase A ballots them into a 32-bit scalar
// mask and phase B's control flow is scalar (s_bitcmp + s_cbranch, no exec-mask divergence).  Half-wave 0 handles the taps
// (nw, ne), half-wave 1 (sw, se).
#pragma once
#include "kplanes_device.h"

#ifdef TN_PHASE_TIMERS
extern __device__ unsigned long long tn_phase_cycles_b[16];
#endif
namespace tn {

constexpr int KP_GS = 36;                                      // floats per tile row: conflict-free b128 writes
constexpr int KP_WAVE_LDS = 32 * KP_GS + 2 * 4 * 32;           // tile + offsets + weights (floats)

// lane (j = sample, h): g[q] = d loss / d feature for channels c0 + CS * q .. + 3 of this scale (q < NV), the channel
// pattern of plane_gather<NV, CS>.  planes / grads: the three [H][W][C] planes of the scale (grads[p] == nullptr: no
// gradient wanted for that plane).  wave_lds: KP_WAVE_LDS floats private to the wave.
template <int NV, int CS>
__device__ __forceinline__ void kp_scatter_scale(const float *const (&planes)[3], float *const (&grads)[3], int H, int W, int C,
                                                 const float (&xs)[3], bool valid, const f32x4k (&g)[NV], int c0,
                                                 float *__restrict__ wave_lds, int j, int h)
{
#ifdef TN_PHASE_TIMERS
    unsigned long long pts_ = __builtin_amdgcn_s_memtime();
#define TN_PTS(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (tn::lane_id() == 0) atomicAdd(&tn_phase_cycles_b[k], n_ - pts_); pts_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
    float *tileG = wave_lds;
    int *tileO = reinterpret_cast<int *>(tileG + 32 * KP_GS);
    float *tileW = tileG + 32 * KP_GS + 4 * 32;
    PlaneTaps t[3];
    f32x4k val[3][NV];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float u, v;
        pair_uv(xs, p, u, v);
        t[p] = plane_taps(u, v, H, W, C);
        if (planes[p]) {
            plane_gather<NV, CS>(planes[p], t[p], c0, val[p]);
            __builtin_amdgcn_sched_barrier(0);      // one plane's 16 loads in flight at a time (register budget)
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) val[p][q] = f32x4k{1.f, 1.f, 1.f, 1.f};
        }
    }
#ifdef TN_PHASE_TIMERS
    TN_PTS(8)
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (grads[p] == nullptr) continue;
        // ---- phase A: lane = (sample j, channel group) ----
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const f32x4k gp = p == 0 ? g[q] * val[1][q] * val[2][q]
                                     : (p == 1 ? g[q] * val[0][q] * val[2][q] : g[q] * (val[0][q] * val[1][q]));
            *reinterpret_cast<f32x4k *>(tileG + j * KP_GS + c0 + CS * q) = gp;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {          // this lane publishes taps 2h, 2h+1 of its sample
            const int o = h ? t[p].off[2 + k] : t[p].off[0 + k];
            const float w = h ? t[p].w[2 + k] : t[p].w[0 + k];
            // a tap outside the plane (or a row past n) gets weight 0 and points at texel 0: its running sum stays exactly 0, so
            // phase B needs no per-atomic bounds test (compare + exec save / restore around every atomic) -- at worst it adds
            // +0.0 to a valid line when a run ends at the plane's border
            const bool live = valid && o >= 0;
            tileO[(2 * h + k) * 32 + j] = live ? 4 * o : 0;            // BYTE offset (planes are < 2^30 elements)
            tileW[(2 * h + k) * 32 + j] = live ? w : 0.0f;
        }
        // run boundaries: sample j closes a run when the next sample falls into another cell.  When the next
        // cell is a 4-neighbour, two of the four texels are shared with it: instead of flushing them, their
        // partial sums are carried into the next run (x moves: within the half-wave; y moves: across halves).
        const int cell = valid ? t[p].cell : -1 - j;
        const int next_cell = __shfl_down(cell, 1, 64);
        const int dcell = (j < 31) ? next_cell - cell : 0x40000000;
        const int rowlen = W + 4;
        const unsigned run_end = (unsigned)__ballot(dcell != 0);                       // low 32 bits: half 0 == half 1
        const unsigned mv_xp = (unsigned)__ballot(dcell == 1), mv_xm = (unsigned)__ballot(dcell == -1);
        const unsigned mv_yp = (unsigned)__ballot(dcell == rowlen), mv_ym = (unsigned)__ballot(dcell == -rowlen);
        // diagonal neighbours share ONE texel with the current cell: three lines are flushed instead of four
        const unsigned mv_pp = (unsigned)__ballot(dcell == rowlen + 1), mv_mp = (unsigned)__ballot(dcell == rowlen - 1);
        const unsigned mv_pm = (unsigned)__ballot(dcell == -rowlen + 1), mv_mm = (unsigned)__ballot(dcell == -rowlen - 1);
        const unsigned mv_diag = mv_pp | mv_mp | mv_pm | mv_mm;
        asm volatile("" ::: "memory");         // DS ops of one wave execute in order; only the compiler must not reorder
#ifdef TN_PHASE_TIMERS
        TN_PTS(9)
#endif
        // ---- phase B: lane = (tap pair h, channel c) ----
        const int c = j;                       // channel
        char *const gplane = reinterpret_cast<char *>(grads[p]);     // wave-uniform base + 32-bit byte offsets: no 64-bit VALU
        const unsigned c4 = 4u * (unsigned)c;                        // address arithmetic per atomic (saddr form)
        auto add_at = [&](unsigned byte_off, float v) { atomicAdd(reinterpret_cast<float *>(gplane + byte_off), v); };
        if (c < C) {
            const int4 *O0 = reinterpret_cast<const int4 *>(tileO + (2 * h) * 32), *O1 = O0 + 8;
            const f32x4k *W0 = reinterpret_cast<const f32x4k *>(tileW + (2 * h) * 32);
            const f32x4k *W1 = W0 + 8;
            float a0 = 0.0f, a1 = 0.0f;        // running sums of this half's left / right texel
            // (not unrolled: fully unrolled, the 32-sample walk with its five-way run logic made the kernel 12 k
            // instructions = 72 KB, more than the instruction cache two CUs share)
#pragma clang loop unroll(disable)
            for (int s4 = 0; s4 < 8; ++s4) {
                const f32x4k w0 = W0[s4], w1 = W1[s4];
                const int4 o0v = O0[s4], o1v = O1[s4];   // offsets of the group up front: a read per run end would expose an LDS latency each
                const int o0a[4] = {o0v.x, o0v.y, o0v.z, o0v.w}, o1a[4] = {o1v.x, o1v.y, o1v.z, o1v.w};
                float gv[4];                   // four samples at a time: 32 at once cost 12 spilled registers
#pragma unroll
                for (int u = 0; u < 4; ++u) gv[u] = tileG[(4 * s4 + u) * KP_GS + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sI = 4 * s4 + u;
                    a0 = fmaf(gv[u], w0[u], a0);
                    a1 = fmaf(gv[u], w1[u], a1);
                    if ((run_end >> sI) & 1u) {          // wave-uniform (scalar) control flow from here on
                        const unsigned o0 = (unsigned)o0a[u] + c4, o1 = (unsigned)o1a[u] + c4;
                        if ((mv_xp >> sI) & 1u) {        // next cell = x+1: right texel becomes the left one
                            add_at(o0, a0);
                            a0 = a1; a1 = 0.0f;
                        } else if ((mv_xm >> sI) & 1u) { // next cell = x-1
                            add_at(o1, a1);
                            a1 = a0; a0 = 0.0f;
                        } else if ((mv_yp >> sI) & 1u) { // next cell = y+1: the lower row (half 1) becomes the upper row
                            const float t0 = __shfl_xor(a0, 32, 64), t1 = __shfl_xor(a1, 32, 64);
                            if (h == 0) {
                                add_at(o0, a0);
                                add_at(o1, a1);
                            }
                            a0 = h == 0 ? t0 : 0.0f; a1 = h == 0 ? t1 : 0.0f;
                        } else if ((mv_ym >> sI) & 1u) { // next cell = y-1
                            const float t0 = __shfl_xor(a0, 32, 64), t1 = __shfl_xor(a1, 32, 64);
                            if (h == 1) {
                                add_at(o0, a0);
                                add_at(o1, a1);
                            }
                            a0 = h == 1 ? t0 : 0.0f; a1 = h == 1 ? t1 : 0.0f;
                        } else if ((mv_diag >> sI) & 1u) {
                            // (dx, dy): the texel diagonally across stays -- upper row = half 0, lower row = half 1, a0 = left, a1 = right
                            const float t0 = __shfl_xor(a0, 32, 64), t1 = __shfl_xor(a1, 32, 64);
                            if ((mv_pp >> sI) & 1u) {            // (+1, +1): lower right -> new upper left
                                add_at(o0, a0);
                                if (h == 0) add_at(o1, a1);
                                a0 = h == 0 ? t1 : 0.0f; a1 = 0.0f;
                            } else if ((mv_mp >> sI) & 1u) {     // (-1, +1): lower left -> new upper right
                                add_at(o1, a1);
                                if (h == 0) add_at(o0, a0);
                                a1 = h == 0 ? t0 : 0.0f; a0 = 0.0f;
                            } else if ((mv_pm >> sI) & 1u) {     // (+1, -1): upper right -> new lower left
                                add_at(o0, a0);
                                if (h == 1) add_at(o1, a1);
                                a0 = h == 1 ? t1 : 0.0f; a1 = 0.0f;
                            } else {                             // (-1, -1): upper left -> new lower right
                                add_at(o1, a1);
                                if (h == 1) add_at(o0, a0);
                                a1 = h == 1 ? t0 : 0.0f; a0 = 0.0f;
                            }
                        } else {
                            add_at(o0, a0);
                            add_at(o1, a1);
                            a0 = 0.0f; a1 = 0.0f;
                        }
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
#ifdef TN_PHASE_TIMERS
        TN_PTS(10)
#endif
    }
}

}  // namespace tn
