// Cobafa factorised feature field (reference src/models.py:209-266): per sample one trilinear lookup into
// the coefficient grid (L channels) and, per level, one trilinear lookup of the sawtooth-warped point into a
// basis grid; features = basis * coefficient, levels concatenated.  Replaces 7 x grid_sampler_3d + transposes
// + muls + cat of the reference with one launch each way.  Channel-last grids: a voxel's channels are one
// 16-32 B run.
#include "tn_common.h"
#include <algorithm>

namespace {

struct CbArgs {
    int n_levels, feat_dim;
    int cres[3];
    int res[TN_COBAFA_MAX_LEVELS][3];
    int ch[TN_COBAFA_MAX_LEVELS], off[TN_COBAFA_MAX_LEVELS];
    float freq[TN_COBAFA_MAX_LEVELS];
    const float *coef;
    const float *basis[TN_COBAFA_MAX_LEVELS];
    float *gcoef;
    float *gbasis[TN_COBAFA_MAX_LEVELS];
};

__device__ __forceinline__ void sawtooth(const float x[3], float f, float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = f * x[c];
        const float s = 2.0f * (v - floorf(v)) - 1.0f;  // torch: (f*x) % 1 in [0,1)  (models.py:213)
        y[c] = f > 0.0f ? s : x[c];                     // f <= 0: plain grid lookup (CobafaGrid on its own)
    }
}

// ATen grid_sampler_3d conventions (align_corners=True, zeros padding); p = (x -> W, y -> H, z -> D)
struct Cell3 {
    int base, mask;                    // voxel index of the (x0,y0,z0) corner; bit k: tap k is inside the grid
    float fx, gx, fy, gy, fz, gz;
};

__device__ __forceinline__ Cell3 cell3(const float p[3], int D, int H, int W) {
    Cell3 t;
    const float ix = ((p[0] + 1.0f) * 0.5f) * (float)(W - 1);
    const float iy = ((p[1] + 1.0f) * 0.5f) * (float)(H - 1);
    const float iz = ((p[2] + 1.0f) * 0.5f) * (float)(D - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    t.fx = ix - x0; t.fy = iy - y0; t.fz = iz - z0;
    t.gx = (x0 + 1.0f) - ix; t.gy = (y0 + 1.0f) - iy; t.gz = (z0 + 1.0f) - iz;
    int m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float cx = x0 + (float)(k & 1), cy = y0 + (float)((k >> 1) & 1), cz = z0 + (float)(k >> 2);
        const bool ok = cx >= 0.0f && cx < (float)W && cy >= 0.0f && cy < (float)H && cz >= 0.0f && cz < (float)D;
        m |= ok ? (1 << k) : 0;
    }
    t.mask = m;
    t.base = m ? ((int)z0 * H + (int)y0) * W + (int)x0 : 0;
    return t;
}
__device__ __forceinline__ float tap_weight(const Cell3 &t, int k) {
    return ((k & 1) ? t.fx : t.gx) * (((k >> 1) & 1) ? t.fy : t.gy) * ((k >> 2) ? t.fz : t.gz);
}
__device__ __forceinline__ int tap_delta(int k, int H, int W) { return (k & 1) + ((k >> 1) & 1) * W + (k >> 2) * W * H; }

__device__ __forceinline__ int rl(int v, int s) { return __builtin_amdgcn_readlane(v, s); }
__device__ __forceinline__ float rl(float v, int s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s)); }

constexpr int CB_WAVES = 4;

// ------------------------------------------------------------------------------------------------
// Both directions in transposed form: lane = (tap k = lane >> 3, channel c = lane & 7), a wave walks its 64 samples.
// A lane-per-sample gather / scatter touches 64 different lines per instruction (the scatter ran at ~20 G lane-atomics/s:
// 17.7 ms for 2^20 samples x 8 taps x 42 channels; this form: 2.2 ms, forward 1.6 -> 0.8 ms); here a level of a
// sample is ONE load instruction whose x-adjacent taps form 32-64 B runs (4 requests), the 8 taps are summed with three
// cross-lane adds, and the backward scatters through the very addresses it gathered from.  Per-sample cells travel as
// wave scalars (v_readlane).  Forward results go through a wave-private LDS tile and leave as one contiguous block;
// the backward stages its slice of grad_feat the same way.  NL > 0: compile-time level count (straight-line level loop).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Cell3 bcast(const Cell3 &t, int s) {
    Cell3 u;
    u.base = rl(t.base, s); u.mask = rl(t.mask, s);
    u.fx = rl(t.fx, s); u.gx = rl(t.gx, s); u.fy = rl(t.fy, s); u.gy = rl(t.gy, s); u.fz = rl(t.fz, s); u.gz = rl(t.gz, s);
    return u;
}
// Cross-lane sums without the LDS crossbar (round 4): __shfl_xor compiles to ds_bpermute_b32 -- an LDS instruction with an
// address operand and a ~60-cycle round trip, 42 of them per sample in the backward -- where the lane layout (tap, channel) only
// ever needs exchanges the VALU can do: DPP within a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {           // v + (v moved by DPP control CTRL), all rows and banks
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rows_add16(float v) {        // lane i += lane i ^ 16 (swap odd rows of one copy with even rows of the other)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_add32(float v) {        // lane i += lane i ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_taps(float v) {          // over k = lane bits 3..5: row_ror:8, then the two row exchanges
    return rows_add32(rows_add16(dpp_add<0x128>(v)));
}
__device__ __forceinline__ float sum_channels(float v) {      // over c = lane bits 0..2: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
    return dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v)));
}

template <bool BWD, int NL>
__global__ __launch_bounds__(CB_WAVES * 64) void cobafa_t_kernel(CbArgs a, const float *__restrict__ x, int64_t n, float *__restrict__ feat,
                                                                 const float *__restrict__ gfeat)
{
    extern __shared__ __attribute__((aligned(16))) float lds_t[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int FD = a.feat_dim, n_levels = NL > 0 ? NL : a.n_levels;
    float *tile = lds_t + wave * 64 * FD;
    const int64_t first = ((int64_t)blockIdx.x * CB_WAVES + wave) * 64;
    if (first >= n) return;                                     // wave-uniform
    const int cnt = (int)(n - first < 64 ? n - first : 64);
    const bool valid = lane < cnt;
    const int64_t i = valid ? first + lane : n - 1;
    const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    // phase 0, lane = sample: the cell of every lookup
    Cell3 tc = cell3(p, a.cres[0], a.cres[1], a.cres[2]);
    Cell3 tl[TN_COBAFA_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
        if (l >= n_levels) break;
        float y[3];
        sawtooth(p, a.freq[l], y);
        tl[l] = cell3(y, a.res[l][0], a.res[l][1], a.res[l][2]);
    }
    if constexpr (BWD) {                                        // this wave's [cnt, FD] block of grad_feat
        const float *src = gfeat + first * FD;
        for (int e = lane; e < cnt * FD; e += 64) tile[e] = src[e];
    }
    const int k = lane >> 3, c = lane & 7;
    const int dkc = tap_delta(k, a.cres[1], a.cres[2]);
    const int cc = c < n_levels ? c : 0;
    // BWD, run merging: consecutive samples of a ray stay in one cell of the coarse / low-frequency lookups for several steps
    // (coefficient grid ~9 samples per cell, levels 0-2: 9 / 4 / 2; levels 3-5 move more than a voxel per step).  The cell of
    // a lookup is a wave scalar, so "same cell as the previous sample" is a scalar branch: the lane's contribution is added
    // to a register and ONE atomic per run goes out when the cell changes (~2.2x fewer atomics per sample).
    float run_c = 0.0f, run_b[TN_COBAFA_MAX_LEVELS];
    int pbase_c = -1, pmask_c = 0, pbase[TN_COBAFA_MAX_LEVELS], pmask[TN_COBAFA_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) { run_b[l] = 0.0f; pbase[l] = -1; pmask[l] = 0; }
    auto flush_basis = [&](int l, int base, int mask, float v) {
        const int C = a.ch[l];
        if (((mask >> k) & 1) && c < C) atomicAdd(a.gbasis[l] + ((int64_t)(base + tap_delta(k, a.res[l][1], a.res[l][2])) * C + c), v);
    };
    auto flush_coef = [&](int base, int mask, float v) {
        if (((mask >> k) & 1) && c < n_levels) atomicAdd(a.gcoef + ((int64_t)(base + dkc) * n_levels + cc), v);
    };
    // Software pipeline over the samples (round 4): the seven lookups of sample s + 1 are requested BEFORE sample s is processed.
    // Written in the obvious order -- load, reduce, scatter, next level -- every load sits behind the previous level's atomics,
    // which hipcc may not reorder it with (the grids and their gradients may alias for all it knows), and a wave paid seven
    // memory round trips per sample one after the other: 5.4 us per sample, 1.83 ms per 2^20 samples with twelve waves per CU.
    float ld_c = 0.0f, ld_b[TN_COBAFA_MAX_LEVELS];
    auto request = [&](int s) {
        const int cbase = rl(tc.base, s), cmask = rl(tc.mask, s);
        const bool okc = ((cmask >> k) & 1) && c < n_levels;
        ld_c = a.coef[(int64_t)(okc ? cbase + dkc : 0) * n_levels + cc];
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
            if (l >= n_levels) break;
            const int C = a.ch[l];
            const int base = rl(tl[l].base, s), mask = rl(tl[l].mask, s);
            const bool ok = ((mask >> k) & 1) && c < C;
            ld_b[l] = a.basis[l][(int64_t)(ok ? base + tap_delta(k, a.res[l][1], a.res[l][2]) : 0) * C + (c < C ? c : 0)];
        }
    };
    request(0);
    for (int s = 0; s < cnt; ++s) {
        const float cur_c = ld_c;
        float cur_b[TN_COBAFA_MAX_LEVELS];
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) cur_b[l] = l < n_levels ? ld_b[l] : 0.0f;
        if (s + 1 < cnt) request(s + 1);                          // (wave-uniform) in flight while sample s is reduced and scattered
        // coefficient lookup: lane c < n_levels ends up with coef[c] of sample s
        const Cell3 uc = bcast(tc, s);
        const bool okc = ((uc.mask >> k) & 1) && c < n_levels;
        const float wc = okc ? tap_weight(uc, k) : 0.0f;
        const float coef = sum_taps(cur_c * wc);
        float gcv = 0.0f;                                       // BWD: lane c = level: d loss / d coef[c] of sample s
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
            if (l >= n_levels) break;
            const int C = a.ch[l];
            const Cell3 u = bcast(tl[l], s);
            const bool ok = ((u.mask >> k) & 1) && c < C;
            const float w = ok ? tap_weight(u, k) : 0.0f;
            const float cl = rl(coef, l);
            float *slot = tile + s * FD + a.off[l] + (c < C ? c : 0);
            if constexpr (!BWD) {
                const float val = sum_taps(cur_b[l] * w);        // basis value of channel c (0 for c >= C)
                if (k == 0 && c < C) *slot = val * cl;
            } else {
                const float g = c < C ? *slot : 0.0f;
                // d feat / d coef[l] = sum_c g_c (sum_k basis_kc w_k): one sum over all 64 lanes (k, c)
                const float gc = sum_taps(sum_channels(g * (cur_b[l] * w)));
                gcv = c == l ? gc : gcv;
                const float v = g * cl * w;                        // d feat / d basis = coef * w  (0 outside the grid: w = 0)
                if (u.base == pbase[l] && u.mask == pmask[l]) run_b[l] += v;
                else {
                    if (pmask[l]) flush_basis(l, pbase[l], pmask[l], run_b[l]);
                    run_b[l] = v; pbase[l] = u.base; pmask[l] = u.mask;
                }
            }
        }
        if constexpr (BWD) {
            const float v = gcv * wc;
            if (uc.base == pbase_c && uc.mask == pmask_c) run_c += v;
            else {
                if (pmask_c) flush_coef(pbase_c, pmask_c, run_c);
                run_c = v; pbase_c = uc.base; pmask_c = uc.mask;
            }
        }
    }
    if constexpr (BWD) {
#pragma unroll
        for (int l = 0; l < TN_COBAFA_MAX_LEVELS; ++l) {
            if (l >= n_levels) break;
            if (pmask[l]) flush_basis(l, pbase[l], pmask[l], run_b[l]);
        }
        if (pmask_c) flush_coef(pbase_c, pmask_c, run_c);
    }
    if constexpr (!BWD) {
        float *dst = feat + first * FD;
        for (int e = lane; e < cnt * FD; e += 64) dst[e] = tile[e];
    }
}

template <bool BWD>
int launch_t(const CbArgs &a, const float *x, int64_t n, float *feat, const float *gfeat, hipStream_t s)
{
    const unsigned blocks = (unsigned)((n + 64 * CB_WAVES - 1) / (64 * CB_WAVES));
    const size_t lds = (size_t)CB_WAVES * 64 * a.feat_dim * sizeof(float);
    if (a.n_levels == 6) cobafa_t_kernel<BWD, 6><<<dim3(blocks), dim3(64 * CB_WAVES), lds, s>>>(a, x, n, feat, gfeat);
    else cobafa_t_kernel<BWD, 0><<<dim3(blocks), dim3(64 * CB_WAVES), lds, s>>>(a, x, n, feat, gfeat);
    return tn::check_launch(BWD ? "cobafa_t_kernel(backward)" : "cobafa_t_kernel(forward)");
}

int make_args(const tn_cobafa_desc *d, CbArgs &a)
{
    TN_REQUIRE(d, TN_E_NULL, "cobafa: null descriptor");
    TN_REQUIRE(d->n_levels >= 1 && d->n_levels <= TN_COBAFA_MAX_LEVELS, TN_E_CONFIG, "cobafa: n_levels out of range");
    TN_REQUIRE(d->coef, TN_E_NULL, "cobafa: null coefficient grid");
    a.n_levels = d->n_levels; a.coef = d->coef; a.gcoef = nullptr;
    int off = 0;
    for (int c = 0; c < 3; ++c) { TN_REQUIRE(d->coef_res[c] > 0, TN_E_SIZE, "cobafa: bad coef resolution"); a.cres[c] = d->coef_res[c]; }
    for (int l = 0; l < d->n_levels; ++l) {
        TN_REQUIRE(d->basis[l], TN_E_NULL, "cobafa: null basis grid");
        TN_REQUIRE(d->channels[l] >= 1 && d->channels[l] <= 8, TN_E_CONFIG, "cobafa: channels must be in [1, 8]");
        for (int c = 0; c < 3; ++c) { TN_REQUIRE(d->res[l][c] > 0, TN_E_SIZE, "cobafa: bad basis resolution"); a.res[l][c] = d->res[l][c]; }
        TN_REQUIRE((int64_t)d->res[l][0] * d->res[l][1] * d->res[l][2] < (1ll << 31), TN_E_SIZE, "cobafa: grid too large");
        a.ch[l] = d->channels[l]; a.off[l] = off; off += d->channels[l];
        a.freq[l] = d->freqs[l]; a.basis[l] = d->basis[l]; a.gbasis[l] = nullptr;
    }
    a.feat_dim = off;
    return TN_OK;
}

}  // namespace

extern "C" int tn_cobafa_fwd(const tn_cobafa_desc *desc, const float *x, int64_t n, float *feat, void *stream)
{
    CbArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_cobafa_fwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && feat, TN_E_NULL, "tn_cobafa_fwd: null pointer");
    return launch_t<false>(a, x, n, feat, nullptr, (hipStream_t)stream);
}

extern "C" int tn_cobafa_bwd(const tn_cobafa_desc *desc, const float *x, int64_t n, const float *grad_feat, float *grad_coef,
                             float *const *grad_basis, void *stream)
{
    CbArgs a;
    if (int rc = make_args(desc, a)) return rc;
    TN_REQUIRE(n >= 0, TN_E_SIZE, "tn_cobafa_bwd: negative n");
    if (n == 0) return TN_OK;
    TN_REQUIRE(x && grad_feat && grad_coef && grad_basis, TN_E_NULL, "tn_cobafa_bwd: null pointer");
    a.gcoef = grad_coef;
    for (int l = 0; l < a.n_levels; ++l) {
        TN_REQUIRE(grad_basis[l], TN_E_NULL, "tn_cobafa_bwd: null basis gradient");
        a.gbasis[l] = grad_basis[l];
    }
    return launch_t<true>(a, x, n, nullptr, grad_feat, (hipStream_t)stream);
}
