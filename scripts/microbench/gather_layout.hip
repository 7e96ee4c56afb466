// TA / L1 cost of gathering 128-byte rows: lane layout A (lane = (sample j, half h), 4 loads of 16 B at 32 q + 16 h: what the K-Planes
// kernels do today) against layout B (8 consecutive lanes read one row's eight 16-byte pieces: 4 loads cover 32 samples) and C
// (B with 32 lanes per row, dword loads: 32 loads per 32 samples... as `full-line` reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void gather(const float *__restrict__ tab, const int *__restrict__ idx, long n_tiles, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (long tile = (long)blockIdx.x * 8 + wave; tile < n_tiles; tile += (long)gridDim.x * 8) {
        const int *ix = idx + tile * 32 * 4;       // 4 taps per sample
        if (MODE == 0) {
            const int j = lane & 31, h = lane >> 5;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *row = tab + (long)ix[4 * j + t] * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) s += *reinterpret_cast<const f32x4 *>(row + 8 * q);
            }
        } else if (MODE == 1) {
            const int c = lane & 7, sl = lane >> 3;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float *row = tab + (long)ix[4 * (8 * g + sl) + t] * 32 + 4 * c;
                    s += *reinterpret_cast<const f32x4 *>(row);
                }
        } else {        // 16 lanes per row, 8-byte loads: 8 loads per tap set
            const int c = lane & 15, sl = lane >> 4;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float *row = tab + (long)ix[4 * (4 * g + sl) + t] * 32 + 2 * c;
                    const float2 v = *reinterpret_cast<const float2 *>(row);
                    s[0] += v.x; s[1] += v.y;
                }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main(int argc, char **argv) {
    const long n = 1 << 20, n_tiles = n / 32;
    for (int res : {128, 512}) {
        const long rows = (long)res * res;
        float *tab; int *idx; float *out;
        hipMalloc(&tab, rows * 128); hipMemset(tab, 0, rows * 128);
        hipMalloc(&idx, n * 4 * 4); hipMalloc(&out, 4096 * 512 * 4);
        std::vector<int> h(n * 4);
        srand(1);
        for (long s = 0; s < n; ++s) {      // bilinear neighbours: (y, x), (y, x+1), (y+1, x), (y+1, x+1); rays: runs of 8 samples share a neighbourhood
            static int by, bx;
            if (s % 8 == 0) { by = rand() % (res - 4); bx = rand() % (res - 4); }
            const int y = by + rand() % 3, x = bx + rand() % 3;
            h[4 * s] = y * res + x; h[4 * s + 1] = y * res + x + 1; h[4 * s + 2] = (y + 1) * res + x; h[4 * s + 3] = (y + 1) * res + x + 1;
        }
        hipMemcpy(idx, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            auto run = [&]() {
                if (mode == 0) gather<0><<<1024, 512>>>(tab, idx, n_tiles, out);
                else if (mode == 1) gather<1><<<1024, 512>>>(tab, idx, n_tiles, out);
                else gather<2><<<1024, 512>>>(tab, idx, n_tiles, out);
            };
            run(); hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); for (int r = 0; r < 10; ++r) run(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
            printf("res %d mode %d: %.1f us for %ld samples x 4 taps x 128 B = %.2f TB/s\n", res, mode, ms * 1e3, n, n * 4 * 128.0 / ms / 1e9);
        }
        hipFree(tab); hipFree(idx); hipFree(out);
    }
    return 0;
}
