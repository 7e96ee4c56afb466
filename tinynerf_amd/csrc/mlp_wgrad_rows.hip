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
};

__device__ __forceinline__ int frow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void glds16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

template <int NG, int NA, int BN, int BK>
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
        const int src_row = row < NG ? row : row - NG;
        src_off[e] = src_row * 32 + 4 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    auto fetch = [&](int64_t tile, float *buf) {
        const float *sg = a.g_rows + tile * a.g_stride, *sa = a.a_rows + tile * a.a_stride;
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const bool is_g = 8 * (wave * NI + e) < NG;                  // wave-uniform
            glds16((is_g ? sg : sa) + src_off[e], buf + 256 * (wave * NI + e));
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 32 * (tn0 + bn) + frow(r, h);
                if (k < a.kmax) atomicAdd(&a.gW[(int64_t)nn * a.ldw + a.col0 + k], acc[bn][bk][r]);
            }
        }
        if (tk0 == 0 && a.gB != nullptr) {
            float s = dbacc[bn];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&a.gB[32 * (tn0 + bn) + i], s);
        }
    }
}

template <int NG, int NA, int BN, int BK>
int launch_rows(const WgradRowsArgs &w, int64_t n, hipStream_t s)
{
    constexpr size_t lds_bytes = (size_t)2 * (NG + NA) * 32 * sizeof(float);
    auto kern = wgrad_rows_kernel<NG, NA, BN, BK>;
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
