// Ablation of fwd_wreg_kernel (mlp_bwd_layers.hip): where do the ~15 % of matrix-pipe idle time go?
// MODE bits: 1 = no global stores, 2 = no global loads / ds_writes (LDS tile reused), 4 = no barriers, 8 = no ds_reads (B operand constant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, T = 8, SW = H + 4, TILE = 32 * SW;
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
template <int MODE>
__global__ __launch_bounds__(512) void layer(const float *__restrict__ Wg, float *__restrict__ stash, int64_t n_tiles, int rows_total)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ob = wave;
    const int64_t stride = gridDim.x, first = blockIdx.x;
    const int64_t iters = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    if (iters == 0) return;
    f32x4 W[T][4];
    const float *wrow = Wg + (int64_t)(32 * ob + j) * H + 4 * h;
    for (int t = 0; t < T; ++t) for (int q = 0; q < 4; ++q) W[t][q] = *reinterpret_cast<const f32x4 *>(wrow + 32 * t + 8 * q);
    auto tile_of = [&](int64_t it) { const int64_t t = first + it * stride; return t < n_tiles ? t : n_tiles - 1; };
    float stage[16];
    auto load = [&](int64_t tile) {
        if (MODE & 256) {
            const char *base = reinterpret_cast<const char *>(stash + (tile * rows_total + 32 * ob) * 32);
            const unsigned off = (unsigned)((16 * h) * 32 + j) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[e] = *reinterpret_cast<const float *>(base + off + (unsigned)(e * 128));
            return;
        }
        if (MODE & 16) {      // same bytes as four 16-byte requests per lane (layout ignored: instruction-count experiment)
            const f32x4 *p = reinterpret_cast<const f32x4 *>(stash + (tile * rows_total) * 32 + 32 * ob * 32) + lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const f32x4 v = p[e * 64]; stage[4 * e] = v[0]; stage[4 * e + 1] = v[1]; stage[4 * e + 2] = v[2]; stage[4 * e + 3] = v[3]; }
            return;
        }
        const float *p = stash + (tile * rows_total) * 32 + (32 * ob + 16 * h) * 32 + j;
#pragma unroll
        for (int e = 0; e < 16; ++e) stage[e] = p[e * 32];
    };
    auto glds = [&](int64_t tile, float *tb) {          // wave ob brings in LDS rows 32 ob .. + 31: four 1-KB LDS-direct loads
        const float *st = stash + (tile * rows_total) * 32;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int R = 32 * ob + 8 * e + (lane >> 3);                    // LDS row <- global row R ^ ((R >> 2) & 1)
            __builtin_amdgcn_global_load_lds(st + (R ^ ((R >> 2) & 1)) * 32 + 4 * (lane & 7),
                                             (__attribute__((address_space(3))) void *)(tb + (32 * ob + 8 * e) * 32), 16, 0, 0);
        }
    };
    float *stg = lds + 3 * TILE + wave * 1024;          // wave-private 32 x 32 staging (MODE & 32)
    auto load4 = [&](int64_t tile) {                    // 32 rows x 32 samples as four 16-byte requests per lane: instr e = rows 8 e .. 8 e + 7
        const f32x4 *p = reinterpret_cast<const f32x4 *>(stash + (tile * rows_total + 32 * ob) * 32) + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const f32x4 v = p[e * 64]; stage[4 * e] = v[0]; stage[4 * e + 1] = v[1]; stage[4 * e + 2] = v[2]; stage[4 * e + 3] = v[3]; }
    };
    auto write4 = [&](float *tb) {                       // lane: row 8 e + (lane >> 3), samples 4 (lane & 7) + c  ->  [sample][feature]
        float *p = tb + (4 * (lane & 7)) * SW + 32 * ob + (lane >> 3);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c * SW + 8 * e] = stage[4 * e + c];
    };
    auto stage_out = [&](const f32x16 &acc) {            // D layout -> staging rows [feature][32 samples]
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
    };
    auto emit4 = [&](int64_t tile) {
        f32x4 *q = reinterpret_cast<f32x4 *>(stash + (tile * rows_total + H + 32 * ob) * 32) + lane;
        const f32x4 *sp = reinterpret_cast<const f32x4 *>(stg) + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e * 64] = sp[e * 64];
    };
    auto write = [&](float *tb) {
        float *p = tb + j * SW + 32 * ob + 16 * h;
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4 *>(p + 4 * v) = f32x4{stage[4 * v], stage[4 * v + 1], stage[4 * v + 2], stage[4 * v + 3]};
    };
    auto emit = [&](int64_t tile, const f32x16 &acc) {
        if (MODE & 256) {
            char *base = reinterpret_cast<char *>(stash + (tile * rows_total + H + 32 * ob) * 32);      // wave-uniform
            const unsigned off = (unsigned)((4 * h) * 32 + j) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<float *>(base + off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128)) = acc[r];
            return;
        }
        if (MODE & 16) {
            f32x4 *p = reinterpret_cast<f32x4 *>(stash + (tile * rows_total + H + 32 * ob) * 32) + lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e * 64] = f32x4{acc[4 * e], acc[4 * e + 1], acc[4 * e + 2], acc[4 * e + 3]};
            return;
        }
        float *outp = stash + (tile * rows_total + H + 32 * ob + 4 * h) * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) outp[((r & 3) + 8 * (r >> 2)) * 32] = acc[r];
    };
    if (MODE & 512) { glds(tile_of(0), lds); }
    else if (MODE & 32) { load4(tile_of(0)); write4(lds); load4(tile_of(1)); } else { load(tile_of(0)); write(lds); load(tile_of(1)); }
    __syncthreads();
    if ((MODE & 64) && wave >= 4) __syncthreads();
    int cur = 0;
    f32x16 res;
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma clang loop unroll(disable)
    for (int64_t it = 0; it < iters; ++it) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        const float *bt = lds + cur * TILE + j * SW + 4 * h;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 1.0f;
        if (MODE & 512) {
            glds(tile_of(it + 1), lds + nxt * TILE);
            if (it > 0) emit(tile_of(it - 1), res);
            __builtin_amdgcn_sched_barrier(0);
            const float *br = lds + cur * TILE + j;
#pragma unroll
            for (int g = 0; g < 4 * T; ++g) {
                if (g == 2 * T) __syncthreads();
                float b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = br[((8 * g + 4 * h + u) ^ h) * 32];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32(W[g >> 2][g & 3][u], b[u], acc);
            }
        } else if (MODE & 128) {
            // memory instructions spread over the first half's MFMAs: pair p (8 MFMAs) is preceded by ds_write p + loads 4p..4p+3 (p < 4)
            // and stores 2p, 2p+1 of the previous tile
            float *wp = lds + nxt * TILE + j * SW + 32 * ob + 16 * h;
            const float *lp = stash + (tile_of(it + 2) * rows_total) * 32 + (32 * ob + 16 * h) * 32 + j;
            float *op = stash + (tile_of(it > 0 ? it - 1 : 0) * rows_total + H + 32 * ob + 4 * h) * 32 + j;
#pragma unroll
            for (int pr = 0; pr < 2 * T; ++pr) {
                if (pr == T) __syncthreads();
                if (pr < 4) {
                    *reinterpret_cast<f32x4 *>(wp + 4 * pr) = f32x4{stage[4 * pr], stage[4 * pr + 1], stage[4 * pr + 2], stage[4 * pr + 3]};
#pragma unroll
                    for (int e = 4 * pr; e < 4 * pr + 4; ++e) stage[e] = lp[e * 32];
                }
                if (pr < 8 && it > 0) {
#pragma unroll
                    for (int r = 2 * pr; r < 2 * pr + 2; ++r) op[((r & 3) + 8 * (r >> 2)) * 32] = res[r];
                }
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bt + 8 * (2 * pr)), b1 = *reinterpret_cast<const f32x4 *>(bt + 8 * (2 * pr + 1));
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32(W[(2 * pr) >> 2][(2 * pr) & 3][u], b0[u], acc);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32(W[(2 * pr + 1) >> 2][(2 * pr + 1) & 3][u], b1[u], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        if (MODE & 32) {
            write4(lds + nxt * TILE);
            if (it > 0) emit4(tile_of(it - 1));
            load4(tile_of(it + 2));
        } else {
        if (!(MODE & 2)) write(lds + nxt * TILE);
        if (!(MODE & 1)) { if (it > 0) emit(tile_of(it - 1), res); }
        if (!(MODE & 2)) load(tile_of(it + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 cb = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
        for (int g = 0; g < 4 * T; ++g) {
            if (g == 2 * T && !(MODE & 4)) __syncthreads();
            f32x4 b = cb;
            if (!(MODE & 8)) b = *reinterpret_cast<const f32x4 *>(bt + 8 * g);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mfma32(W[g >> 2][g & 3][u], b[u], acc);
        }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(acc[r])); res[r] = fmaxf(acc[r], 0.f); }
        if (MODE & 64) __syncthreads();
        if (MODE & 32) stage_out(res);
        cur = nxt;
    }
    if ((MODE & 64) && wave < 4) __syncthreads();
    if (MODE & 32) emit4(tile_of(iters - 1)); else emit(tile_of(iters - 1), res);
}
template <int MODE> void run(const float *W, float *stash, int64_t n_tiles) {
    const size_t ldsb = 3 * TILE * 4 + 8 * 4096;
    hipFuncSetAttribute((const void *)layer<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a); layer<MODE><<<256, 512, ldsb>>>(W, stash, n_tiles, 2 * H); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double mf = (double)n_tiles * 8 * 128;
    printf("mode %3d (%s%s%s%s%s%s%s%s%s%s): %.3f ms, %.1f cycles per MFMA and SIMD, pipe %.1f %% busy\n", MODE, MODE & 1 ? "no-stores " : "", MODE & 2 ? "no-loads " : "",
           MODE & 4 ? "no-barrier " : "", MODE & 8 ? "no-ds_read " : "", MODE & 16 ? "wide-vmem " : "", MODE & 32 ? "wide-vmem+LDS-transposes " : "", MODE & 64 ? "staggered " : "", MODE & 128 ? "interleaved " : "", MODE & 256 ? "saddr " : "", MODE & 512 ? "glds+b32 " : "", best, best * 1e-3 * 2.4e9 / (mf / 1024), 64.0 / (best * 1e-3 * 2.4e9 / (mf / 1024)) * 100);
}
int main() {
    const int64_t n_tiles = 36352;      // 1.16 M samples
    float *W, *stash;
    hipMalloc(&W, H * H * 4); hipMalloc(&stash, n_tiles * 2 * H * 32 * 4);
    hipMemset(W, 0, H * H * 4); hipMemset(stash, 0, n_tiles * 2 * H * 32 * 4);
    run<0>(W, stash, n_tiles); run<1>(W, stash, n_tiles); run<2>(W, stash, n_tiles); run<3>(W, stash, n_tiles); run<4>(W, stash, n_tiles);
    run<7>(W, stash, n_tiles); run<16>(W, stash, n_tiles); run<256>(W, stash, n_tiles); run<512>(W, stash, n_tiles); run<768>(W, stash, n_tiles);
    return 0;
}
