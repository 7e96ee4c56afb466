// Weight gradient of a head's FIRST layer over its x columns, with x given as [feature][32-sample] rows.
//
// The heads behind a wide feature stack (reference models.py:70-89 on VanillaFeatureMLP(10, 256, 8), run.py:131-134; Cobafa's
// 128-wide stack, run.py:141-150) read a [n, 256] feature tensor.  The general two-pass weight-gradient kernel
// (mlp_bwd2.hip) stages the tile's x rows sample-major in LDS and reads the first layer's A-side operand column-wise out of
// them: 20 of its 34 accumulator tiles, 48 slots on 16 waves with a 128-register budget -- 60 spilled registers, 1.9 ms per
// step at 24 % of the matrix rate.  The feature stack's last launch leaves the same values as [feature][32 samples] rows in
// ITS workspace (tn_mlp_rows_view): the operand layout of a sample-reducing MFMA tile.  This kernel is the layer-kernel form
// of mlp_bwd_layers.hip (wgrad_lds_kernel) with two row sources:
//
//     dW_0[n][col0 + k] += sum_s G_0[n][s] * X^T[k][s]        G_0: NG = 64 rows of the head's workspace, X^T: NA rows
//     db_0[n]           += sum_s G_0[n][s]
//
// A workgroup brings a tile's NG + NA rows in once, LDS-direct (global_load_lds_dwordx4, double-buffered, one barrier per
// tile, chunk-swizzled through the source address exactly as in wgrad_lds_kernel), a wave owns BN x BK output tiles for the
// whole launch and flushes them with full-line atomics.  2 * 64 * NA FLOP against 4 * (64 + NA) * 32 bytes per tile:
// 25.6 FLOP/B at NA = 256 -- on the ridge of the fp32-MFMA / HBM roofline (157 TFLOP/s : 6.3 TB/s = 25).
#include "mlp_stage.h"
#include "b3_device.h"
#include <algorithm>

namespace {

using tn::f32x16;
using tn::f32x4;

struct WgradRowsArgs {
    const float *g_rows;        // G_0 rows of tile t at g_rows + t * g_stride   ([NG][32] floats)
    const float *a_rows;        // X^T rows of tile t at a_rows + t * a_stride   ([NA][32] floats)
    int64_t g_stride, a_stride; // floats
    float *gW;                  // [NG][ldw]: dW[n][col0 + k]
    float *gB;                  // [NG] or nullptr
    int ldw, col0, kmax;        // columns k >= kmax are padding (not written)
    // two heads on the same x (g_rows2 != nullptr): G rows [0, NG / 2) come from g_rows and belong to gW / gB, rows [NG / 2, NG) from
    // g_rows2 and belong to gW2 / gB2 -- the x rows, three quarters of a tile's bytes at 64 + 64 against 256 rows, are read once
    const float *g_rows2 = nullptr;
    int64_t g_stride2 = 0;
    float *gW2 = nullptr, *gB2 = nullptr;
    int ldw2 = 0, col02 = 0;
};

__device__ __forceinline__ int frow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void glds16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

// B3: the products as exact bf16 triplets (b3_device.h: six 32 x 32 x 16 MFMAs per 16 samples instead of eight 32 x 32 x 2 fp32 ones per
// 16 -- 2.7 x less matrix time, results equal to fp32 rounding): the two-head form owns four tiles per wave and is matrix-bound on the fp32
// MFMA (4 096 pipe cycles per tile against 2.1 us of HBM time).
template <int NG, int NA, int BN, int BK, bool B3 = false>
__global__ __launch_bounds__(512) void wgrad_rows_kernel(WgradRowsArgs a, int64_t n)
{
    constexpr int NR = NG + NA;                           // rows per tile buffer
    constexpr int TILE = NR * 32;                         // floats per buffer
    constexpr int NI = NR / 64;                           // LDS-direct loads per wave and tile (8 rows each, 8 waves)
    constexpr int WK = (NA / 32) / BK;                    // waves along k
    static_assert(NR % 64 == 0 && NG % 8 == 0 && (NG / 32) % BN == 0 && (NA / 32) % BK == 0 && (NG / 32 / BN) * WK == 8, "8 waves own all tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = tn::lane_id(), i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_tiles = (n + 31) >> 5;
    const int tn0 = (wave / WK) * BN, tk0 = (wave % WK) * BK;
    f32x16 acc[BN][BK];
    float dbacc[BN];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) {
        dbacc[bn] = 0.0f;
#pragma unroll
        for (int bk = 0; bk < BK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bn][bk][r] = 0.0f;
    }
    // this lane's part of a tile fetch: 8-row block b = wave * NI + e (wave-uniform: one source per instruction), row 8 b +
    // (lane >> 3), chunk slot lane & 7 holding chunk (lane & 7) ^ ((row >> 1) & 7)
    int src_off[NI];
#pragma unroll
    for (int e = 0; e < NI; ++e) {
        const int row = 8 * (wave * NI + e) + (lane >> 3);
        const int src_row = row < NG ? ((a.g_rows2 != nullptr && row >= NG / 2) ? row - NG / 2 : row) : row - NG;
        src_off[e] = src_row * 32 + 4 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    auto fetch = [&](int64_t tile, float *buf) {
        const float *sg = a.g_rows + tile * a.g_stride, *sa = a.a_rows + tile * a.a_stride;
        const float *sg2 = a.g_rows2 != nullptr ? a.g_rows2 + tile * a.g_stride2 : sg;
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const int r0 = 8 * (wave * NI + e);                          // wave-uniform: one source per instruction
            const float *src = r0 < NG ? ((a.g_rows2 != nullptr && r0 >= NG / 2) ? sg2 : sg) : sa;
            glds16(src + src_off[e], buf + 256 * (wave * NI + e));
        }
    };
    int g_off[BN], a_off[BK];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) g_off[bn] = (32 * (tn0 + bn) + i) * 32;
#pragma unroll
    for (int bk = 0; bk < BK; ++bk) a_off[bk] = (NG + 32 * (tk0 + bk) + i) * 32;
    const int swz = (i >> 1) & 7;                          // rows 32 b + i (NG % 32 == 0): (r >> 1) & 7 == (i >> 1) & 7
    if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x, lds);
    __syncthreads();
    int cur = 0;
#pragma clang loop unroll(disable)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const float *buf = lds + cur * TILE;
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x, lds + (cur ^ 1) * TILE);
        f32x4 gv[BN][4];
#pragma unroll
        for (int bn = 0; bn < BN; ++bn)
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[bn][e] = *reinterpret_cast<const f32x4 *>(buf + g_off[bn] + 4 * ((4 * h + e) ^ swz));
        if constexpr (B3) {
            // k block sb of the 32 x 32 x 16 products: lane (i, h) supplies samples 16 h + 8 sb .. + 7 of its row (chunks 2 sb, 2 sb + 1 of
            // its half) -- the same eight samples on both operands, every sample of the tile in exactly one (h, sb)
            tn::b3::Op gt[BN][2];
#pragma unroll
            for (int bn = 0; bn < BN; ++bn)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const float v[8] = {gv[bn][2 * sb][0], gv[bn][2 * sb][1], gv[bn][2 * sb][2], gv[bn][2 * sb][3],
                                        gv[bn][2 * sb + 1][0], gv[bn][2 * sb + 1][1], gv[bn][2 * sb + 1][2], gv[bn][2 * sb + 1][3]};
                    gt[bn][sb] = tn::b3::split8(v);
                }
#pragma unroll
            for (int bk = 0; bk < BK; ++bk) {
                f32x4 av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = *reinterpret_cast<const f32x4 *>(buf + a_off[bk] + 4 * ((4 * h + e) ^ swz));
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const float v[8] = {av[2 * sb][0], av[2 * sb][1], av[2 * sb][2], av[2 * sb][3], av[2 * sb + 1][0], av[2 * sb + 1][1], av[2 * sb + 1][2], av[2 * sb + 1][3]};
                    const tn::b3::Op at = tn::b3::split8(v);
#pragma unroll
                    for (int bn = 0; bn < BN; ++bn) acc[bn][bk] = tn::b3::mfma6(gt[bn][sb], at, acc[bn][bk]);
                }
            }
        } else {
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            f32x4 av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = *reinterpret_cast<const f32x4 *>(buf + a_off[bk] + 4 * ((4 * h + e) ^ swz));
#pragma unroll
            for (int bn = 0; bn < BN; ++bn)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[bn][bk] = tn::mfma32(gv[bn][e][u], av[e][u], acc[bn][bk]);
        }
        }
        if (tk0 == 0) {
#pragma unroll
            for (int bn = 0; bn < BN; ++bn) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s += (gv[bn][e][0] + gv[bn][e][1]) + (gv[bn][e][2] + gv[bn][e][3]);
                dbacc[bn] += s;
            }
        }
        __syncthreads();                                   // (drains this wave's LDS-direct loads: the next tile is in place)
        cur ^= 1;
    }
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) {
#pragma unroll
        for (int bk = 0; bk < BK; ++bk) {
            tn::pin16(acc[bn][bk]);
            const int k = 32 * (tk0 + bk) + i;
            const bool second = a.g_rows2 != nullptr && 32 * (tn0 + bn) >= NG / 2;       // (wave-uniform: NG / 2 is a multiple of 32)
            float *gw = second ? a.gW2 : a.gW;
            const int ldw = second ? a.ldw2 : a.ldw, col0 = second ? a.col02 : a.col0, n0 = second ? NG / 2 : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * (tn0 + bn) + frow(r, h) - n0;
                if (k < a.kmax) atomicAdd(&gw[(int64_t)nn * ldw + col0 + k], acc[bn][bk][r]);
            }
        }
        const bool second = a.g_rows2 != nullptr && 32 * (tn0 + bn) >= NG / 2;
        float *gb = second ? a.gB2 : a.gB;
        if (tk0 == 0 && gb != nullptr) {
            float s = dbacc[bn];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&gb[32 * (tn0 + bn) + i - (second ? NG / 2 : 0)], s);
        }
    }
}

template <int NG, int NA, int BN, int BK, bool B3 = false>
int launch_rows(const WgradRowsArgs &w, int64_t n, hipStream_t s)
{
    constexpr size_t lds_bytes = (size_t)2 * (NG + NA) * 32 * sizeof(float);
    auto kern = wgrad_rows_kernel<NG, NA, BN, BK, B3>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { tn::set_error("mlp_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    const int64_t n_tiles = (n + 31) / 32;
    const int per_cu = lds_bytes * 2 <= (size_t)tn::mlp::LDS_LIMIT_BYTES ? 2 : 1;
    kern<<<dim3((unsigned)std::min<int64_t>(n_tiles, 256 * per_cu)), dim3(512), lds_bytes, s>>>(w, n);
    return tn::check_launch("wgrad_rows_kernel");
}

}  // namespace

// internal entry point (mlp_bwd2.hip, mlp_bwd_layers.hip): dW[NG][col0 + k] += G[NG][s] A[NA][s]^T for k < kmax, db[NG] += sum_s G
//   (64, 256) / (64, 128): first layer of a width-64 head over the x columns it reads from a 256- / 128-wide stack;
//   (256, 64) / (128, 64): first layer of a wide stack itself over its <= 64 input slots staged as rows of its workspace (the
//   positional-encoding slots of the Vanilla stack; Cobafa's 36 plain features, zero padded)
extern "C" __attribute__((visibility("hidden"))) int tn_mlp_wgrad_rows(const float *g_rows, int64_t g_stride, int ng, const float *a_rows,
                                                                      int64_t a_stride, int na, float *gW, int ldw, int col0, int kmax,
                                                                      float *gB, int64_t n, void *stream)
{
    WgradRowsArgs w;
    w.g_rows = g_rows; w.a_rows = a_rows; w.g_stride = g_stride; w.a_stride = a_stride; w.gW = gW; w.gB = gB; w.ldw = ldw; w.col0 = col0;
    w.kmax = kmax;
    hipStream_t s = (hipStream_t)stream;
    if (ng == 64 && na == 256) return launch_rows<64, 256, 2, 1>(w, n, s);       // 2 x 8 tiles: wave = one k block, both row blocks
    if (ng == 64 && na == 128) return launch_rows<64, 128, 1, 1>(w, n, s);       // 2 x 4 tiles
    if (ng == 256 && na == 64) return launch_rows<256, 64, 2, 1>(w, n, s);       // 8 x 2 tiles: wave = two row blocks, one k block
    if (ng == 128 && na == 64) return launch_rows<128, 64, 1, 1>(w, n, s);       // 4 x 2 tiles: one per wave
    return tn::fail(TN_E_CONFIG, "mlp_bwd: row-operand weight gradient is built for 64 x 256, 64 x 128, 256 x 64 and 128 x 64");
}

// ... of TWO width-64 heads on the same x rows in one launch (the heads behind a 256- / 128-wide stack: x is 80 % / 67 % of what a
// single-head launch reads)
extern "C" __attribute__((visibility("hidden"))) int tn_mlp_wgrad_rows2(const float *g_rows, int64_t g_stride, const float *g_rows2, int64_t g_stride2,
                                                                       const float *a_rows, int64_t a_stride, int na, float *gW, int ldw, int col0,
                                                                       float *gB, float *gW2, int ldw2, int col02, float *gB2, int64_t n, void *stream)
{
    WgradRowsArgs w;
    w.g_rows = g_rows; w.a_rows = a_rows; w.g_stride = g_stride; w.a_stride = a_stride; w.gW = gW; w.gB = gB; w.ldw = ldw; w.col0 = col0;
    w.kmax = na;
    w.g_rows2 = g_rows2; w.g_stride2 = g_stride2; w.gW2 = gW2; w.gB2 = gB2; w.ldw2 = ldw2; w.col02 = col02;
    hipStream_t s = (hipStream_t)stream;
    if (na == 256) return launch_rows<128, 256, 2, 2, true>(w, n, s);       // 4 x 8 tiles: wave = two row blocks x two k blocks
    if (na == 128) return launch_rows<128, 128, 2, 1, true>(w, n, s);       // 4 x 4 tiles: wave = two row blocks, one k block
    return tn::fail(TN_E_CONFIG, "mlp_bwd: the two-head row-operand weight gradient is built for 128 or 256 x columns");
}
