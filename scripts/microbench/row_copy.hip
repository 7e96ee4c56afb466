// What can a layer launch of the wide stacks reach?  It reads 32 rows x 128 B x (H / 32) = H x 128 B per 32-sample tile at one offset of
// the tile's workspace and writes as much at another (forward / data gradient), or reads twice that (weight gradient).  This copies /
// reads exactly that pattern with nothing else in the kernel: float4 or dword requests, nt hints, 1 .. 4 tiles in flight per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: f4 load, f4 store   1: f4 load, dword stores (two full lines per instruction, wreg_store_block)   2: f4 nt load + nt store   3: read only (two row sets)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void tile_copy(const float *__restrict__ ws, float *__restrict__ wso, long n_tiles, long rows_total, int off_in, int off_out, float *sink)
{
    constexpr int H = 256;                       // rows per tile
    const int t = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (long tile0 = blockIdx.x; tile0 < n_tiles; tile0 += (long)gridDim.x * DEPTH) {
        f4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long tile = tile0 + (long)d * gridDim.x < n_tiles ? tile0 + (long)d * gridDim.x : tile0;
            const f4 *src = reinterpret_cast<const f4 *>(ws + (tile * rows_total + off_in) * 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 2) v[d][c] = __builtin_nontemporal_load(src + t + 512 * c);
                else v[d][c] = src[t + 512 * c];
            }
            if (MODE == 3) {
                const f4 *src2 = reinterpret_cast<const f4 *>(ws + (tile * rows_total + off_out) * 32);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc += src2[t + 512 * c];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long tile = tile0 + (long)d * gridDim.x;
            if (tile >= n_tiles) break;
            float *dst = wso + (tile * rows_total + off_out) * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 0) reinterpret_cast<f4 *>(dst)[t + 512 * c] = v[d][c];
                else if (MODE == 2) __builtin_nontemporal_store(v[d][c], reinterpret_cast<f4 *>(dst) + t + 512 * c);
                else if (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[(c * 4 + e) * 512 + t] = v[d][c][e];
                } else acc += v[d][c];
            }
        }
    }
    if (MODE == 3 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = 1.f;
}
template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const long n = 1 << 20, n_tiles = n / 32;
    for (long rows_total : {512L, 3000L}) {
        float *ws, *sink; hipMalloc(&sink, 64);
        const size_t bytes = (size_t)n_tiles * rows_total * 128;
        hipMalloc(&ws, bytes); hipMemset(ws, 0, bytes);
        const int off_in = 0, off_out = 256;
        const double gb = (double)n_tiles * 256 * 128 / 1e9;
        printf("rows_total %ld (workspace %.1f GB), %.2f GB read + %.2f GB written per launch\n", rows_total, bytes / 1e9, gb, gb);
#define RUN(MODE, DEPTH, BL, what) { float ms = timeit([&] { tile_copy<MODE, DEPTH><<<BL, 512>>>(ws, ws, n_tiles, rows_total, off_in, off_out, sink); }); \
        printf("  %-34s depth %d, %4d workgroups: %.3f ms = %.2f TB/s\n", what, DEPTH, BL, ms, 2 * gb / ms); }
        for (int bl : {256, 512, 1024}) {
            if (bl == 256) { RUN(0, 1, 256, "copy f4 / f4") RUN(0, 2, 256, "copy f4 / f4") RUN(0, 4, 256, "copy f4 / f4") RUN(1, 2, 256, "copy f4 / dword lines") RUN(2, 2, 256, "copy nt f4 / nt f4") RUN(3, 1, 256, "read two row sets") RUN(3, 2, 256, "read two row sets") }
            if (bl == 512) { RUN(0, 1, 512, "copy f4 / f4") RUN(0, 2, 512, "copy f4 / f4") RUN(1, 2, 512, "copy f4 / dword lines") RUN(2, 2, 512, "copy nt f4 / nt f4") RUN(3, 1, 512, "read two row sets") RUN(3, 2, 512, "read two row sets") }
            if (bl == 1024) { RUN(0, 1, 1024, "copy f4 / f4") RUN(0, 2, 1024, "copy f4 / f4") RUN(2, 1, 1024, "copy nt f4 / nt f4") RUN(3, 1, 1024, "read two row sets") RUN(3, 2, 1024, "read two row sets") }
        }
        hipFree(ws); hipFree(sink);
    }
    {   // slab layout: every row set contiguous over the tiles ([row set][tile][256 rows]) -- source and destination are separate 1 GB arrays
        float *a, *b, *sink; hipMalloc(&sink, 64);
        const size_t bytes = (size_t)n_tiles * 256 * 128;
        hipMalloc(&a, 11 * bytes); b = a + 7 * (bytes / 4); hipMemset(a, 0, 11 * bytes);
        const double gb = bytes / 1e9;
        printf("slab layout (source and destination contiguous, 7 slabs apart)\n");
#define RUNS(MODE, DEPTH, BL, what) { float ms = timeit([&] { tile_copy<MODE, DEPTH><<<BL, 512>>>(a, b, n_tiles, 256, 0, 0, sink); }); \
        printf("  %-34s depth %d, %4d workgroups: %.3f ms = %.2f TB/s\n", what, DEPTH, BL, ms, 2 * gb / ms); }
        RUNS(0, 1, 256, "copy f4 / f4") RUNS(0, 2, 256, "copy f4 / f4") RUNS(1, 2, 256, "copy f4 / dword lines") RUNS(2, 2, 256, "copy nt f4 / nt f4")
        RUNS(0, 1, 512, "copy f4 / f4") RUNS(0, 2, 512, "copy f4 / f4") RUNS(2, 2, 512, "copy nt f4 / nt f4") RUNS(2, 1, 1024, "copy nt f4 / nt f4")
    }
    return 0;
}
